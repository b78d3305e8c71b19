#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# upper bound of removing the per-batch clear of the per-bounce masks (655 MB per batch): NOCLR = timing build without the clear (wrong frames); same box, interleaved, 128 spp"
  echo "# icosphere"; bash scripts/dev_ab_frame.sh NOCLR
  echo "# clustered"; MESH=clustered bash scripts/dev_ab_frame.sh NOCLR; } 2>&1 | tee gpurun_out/r05/ab_no_mask_clear.txt
