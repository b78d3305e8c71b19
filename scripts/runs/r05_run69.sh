#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout 1800 python3 -m pytest tests -m gpu -x -q --durations=25 2>&1 | tail -45 | tee gpurun_out/r05/gpu_suite_durations.txt; nproc
