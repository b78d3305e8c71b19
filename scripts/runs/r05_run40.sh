#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# k_spatial_resolve: neighbours prefetched all at once (base: 164 VGPRs, three waves per SIMD) against in groups of 3 / 2 / 1 (145 / 130 / see below VGPRs); frames at 128 spp"
  echo "# icosphere"; bash scripts/dev_ab_frame.sh SG3 SG2 SG1
  echo "# clustered"; MESH=clustered bash scripts/dev_ab_frame.sh SG3 SG2 SG1; } 2>&1 | tee gpurun_out/r05/ab_resolve_groups.txt
