#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# kernel alone: base = mask + per-wave stealing stride, GRAB0 = blind, step 1 (rounds 1-4), G0S = blind + stride, G3N = mask, step 1"
  bash scripts/dev_ab.sh 0 GRAB0 G0S G3N
  echo "## wave times, base"; python3 scripts/dev_wave_times.py 7 2>&1 | grep -v amdgpu.ids
  echo "## wave times, G0S"; MIRRES_LIB=$PWD/ab/libmirres_G0S.so python3 scripts/dev_wave_times.py 7 2>&1 | grep -v amdgpu.ids
  echo "# background-only strip / strip 4 of 8 (256 spp)"
  for v in GRAB0 G0S; do for cfg in "8 4 256 2 bg" "8 4 256 2"; do MIRRES_LIB=$PWD/ab/libmirres_$v.so python3 scripts/dev_strip_one.py $cfg 2>&1 | tail -1 | sed "s/^/$v /"; done; done
  for cfg in "8 4 256 2 bg" "8 4 256 2"; do python3 scripts/dev_strip_one.py $cfg 2>&1 | tail -1 | sed "s/^/base /"; done
  echo "# frames, icosphere"; bash scripts/dev_ab_frame.sh GRAB0 G0S
} > gpurun_out/r05/ab_grab_stride.txt 2>&1
cat gpurun_out/r05/ab_grab_stride.txt
