#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# ordered closest-hit kernel alone, icosphere: base = pushes without range checks while three entries fit the LDS part, NOFP = checked pushes"; bash scripts/dev_ab.sh 2 NOFP
  echo "# lego-like"; MIRRES_MESH=clustered bash scripts/dev_ab.sh 2 NOFP; } > gpurun_out/r05/ab_closest_fastpush.txt 2>&1
cat gpurun_out/r05/ab_closest_fastpush.txt
