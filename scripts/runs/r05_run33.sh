#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# k_new_dir_gen: no read of the path state at bounce 0, no write for background slots in the batched (live-list) mode (base) against the previous build (PREV); frames at 128 spp"
  echo "# icosphere"; bash scripts/dev_ab_frame.sh PREV
  echo "# clustered"; MESH=clustered bash scripts/dev_ab_frame.sh PREV; } > gpurun_out/r05/ab_new_dir_gen.txt 2>&1
cat gpurun_out/r05/ab_new_dir_gen.txt
timeout -k 5 300 python3 scripts/dev_mask_reuse_check.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 5 1200 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_fullsize.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -3
