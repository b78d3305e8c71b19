#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout 1200 python3 -m pytest tests/test_gpu_rccl.py tests/test_gpu_render.py tests/test_gpu_raster.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r05/rccl_tests.txt
