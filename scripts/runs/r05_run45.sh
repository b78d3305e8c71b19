#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_raster.py tests/test_gpu_bvh.py -m gpu -x -q 2>&1 | tail -30 | tee gpurun_out/r05/raster_tests.txt
