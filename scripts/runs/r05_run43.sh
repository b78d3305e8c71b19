#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# shadow-ray kernel: the nearest passing child first (base) against the first passing child first (NOSORT: no entry-distance compare / swap per child); kernel alone, icosphere"; bash scripts/dev_ab.sh 0 NOSORT
  echo "# lego-like"; MIRRES_MESH=clustered bash scripts/dev_ab.sh 0 NOSORT
  echo "# frames, icosphere"; bash scripts/dev_ab_frame.sh NOSORT
  echo "# frames, clustered"; MESH=clustered bash scripts/dev_ab_frame.sh NOSORT; } 2>&1 | tee gpurun_out/r05/ab_any_nosort.txt
