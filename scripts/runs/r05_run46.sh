#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ timeout 300 python3 scripts/dev_raster_time.py; MIRRES_MESH=clustered timeout 300 python3 scripts/dev_raster_time.py; } 2>&1 | grep -v Warning | tee gpurun_out/r05/raster_time.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r05/gpu_suite.txt
