#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# k_spatial_resolve: paired hit loads + the pixel record kept for the fused temporal merge (base) against the previous build (PREV); frames at 128 spp"
  echo "# icosphere"; bash scripts/dev_ab_frame.sh PREV
  echo "# clustered"; MESH=clustered bash scripts/dev_ab_frame.sh PREV; } > gpurun_out/r05/ab_resolve_hit2.txt 2>&1
cat gpurun_out/r05/ab_resolve_hit2.txt
timeout -k 5 900 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_passes.py -m gpu -x -q 2>&1 | tail -3
