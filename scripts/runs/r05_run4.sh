#!/bin/bash
# Round 5, GPU run 4: queue-head probing modes (0 blind atomic, 1 load first always, 2 = default: blind until the first failure) — kernel alone and frames
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# shadow-ray kernel alone (scripts/dev_any_pmc.py 1600 7 10): base = MR_GRAB_MODE 2"; bash scripts/dev_ab.sh 0 GRAB0 GRAB1
  echo "# frames, icosphere"; bash scripts/dev_ab_frame.sh GRAB0 GRAB1
  echo "# frames, clustered"; MESH=clustered bash scripts/dev_ab_frame.sh GRAB0 GRAB1
  echo "# one strip of eight / background only / whole frame (256 spp)"
  for cfg in "8 4 256 2 bg" "8 4 256 2" "1 0 256 2"; do python3 scripts/dev_strip_one.py $cfg 2>&1 | tail -1; MIRRES_LIB=$PWD/ab/libmirres_GRAB0.so python3 scripts/dev_strip_one.py $cfg 2>&1 | tail -1 | sed 's/^/GRAB0 /'; done
} > gpurun_out/r05/ab_grab_modes.txt 2>&1
cat gpurun_out/r05/ab_grab_modes.txt
