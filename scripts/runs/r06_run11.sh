#!/bin/bash
# round 6, run 11: the whole GPU suite on the current sources
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
timeout 2400 python3 -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -35 | tee gpurun_out/r06/gpu_suite.txt
