#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# s_setprio 1 / 3 at the start of k_spatial_gen and k_spatial_resolve (PR1 / PR3) against none (base); frames at 128 spp"
  echo "# icosphere"; bash scripts/dev_ab_frame.sh PR1 PR3
  echo "# clustered"; MESH=clustered bash scripts/dev_ab_frame.sh PR1 PR3; } 2>&1 | tee gpurun_out/r05/ab_chain_prio.txt
