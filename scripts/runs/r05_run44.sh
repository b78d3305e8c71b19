#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# shadow-ray kernel, which passing child first: nearest (base), longest stretch inside the box (ORD1), leaves before nodes then nearest (ORD2); kernel alone, icosphere"; bash scripts/dev_ab.sh 0 ORD1 ORD2
  echo "# lego-like"; MIRRES_MESH=clustered bash scripts/dev_ab.sh 0 ORD1 ORD2; } 2>&1 | tee gpurun_out/r05/ab_any_order.txt
