#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
for i in 1 2 3; do timeout -k 5 200 python3 -m pytest tests/test_gpu_render.py -m gpu -x -q -k "strip" 2>&1 | tail -2; echo "rc=${PIPESTATUS[0]}"; done
