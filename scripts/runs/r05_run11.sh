#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# kernel alone: base = mask of empty sub-queues published once per sub-queue, GRAB0 = blind (rounds 1-4)"
  bash scripts/dev_ab.sh 0 GRAB0
  echo "## wave times, base"; python3 scripts/dev_wave_times.py 7 2>&1 | grep -v amdgpu.ids
  echo "# background-only strip / strip 4 of 8 / whole frame (256 spp)"
  for cfg in "8 4 256 2 bg" "8 4 256 2" "1 0 256 2"; do python3 scripts/dev_strip_one.py $cfg 2>&1 | tail -1 | sed "s/^/base /"; MIRRES_LIB=$PWD/ab/libmirres_GRAB0.so python3 scripts/dev_strip_one.py $cfg 2>&1 | tail -1 | sed "s/^/GRAB0 /"; done
  echo "# frames, icosphere"; bash scripts/dev_ab_frame.sh GRAB0
  echo "# frames, clustered"; MESH=clustered bash scripts/dev_ab_frame.sh GRAB0
} > gpurun_out/r05/ab_grab_once.txt 2>&1
cat gpurun_out/r05/ab_grab_once.txt
