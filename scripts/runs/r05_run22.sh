#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# ordered closest-hit kernel alone, icosphere: refill threshold (lanes still busy below which idle lanes take new rays); base = 40"; bash scripts/dev_ab.sh 2 CR28 CR34 CR48 CR56
  echo "# lego-like"; MIRRES_MESH=clustered bash scripts/dev_ab.sh 2 CR28 CR34 CR48 CR56; } > gpurun_out/r05/ab_closest_refill.txt 2>&1
cat gpurun_out/r05/ab_closest_refill.txt
