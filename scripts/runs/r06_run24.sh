#!/bin/bash
# round 6, run 24: the whole GPU suite on the final sources (closest-hit pushes / shadow-ray pop as straight-line code), then the collection
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06/gpu_suite_c.txt
timeout 3000 bash scripts/profile_r06.sh 2>&1 | tail -30
