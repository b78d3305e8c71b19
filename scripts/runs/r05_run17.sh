#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout -k 5 1800 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_rccl.py tests/test_gpu_raster.py -m gpu -x -q > gpurun_out/r05/gputests_run17.txt 2>&1; tail -4 gpurun_out/r05/gputests_run17.txt
grep -n "AssertionError" -A3 gpurun_out/r05/gputests_run17.txt | head
