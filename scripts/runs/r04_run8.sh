#!/bin/bash
# Round 4, GPU run 8: the whole -m gpu suite on the final kernels, then the round's profile collection
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04 gpurun_out/out
export MIRRES_PARITY_REPORT=$PWD/gpurun_out/r04/parity_report.txt; rm -f $MIRRES_PARITY_REPORT gpurun_out/clustered_mesh_report.txt
( time timeout 2400 python3 -m pytest tests -m gpu -q ) > gpurun_out/r04/gpu_tests.log 2>&1
unset MIRRES_PARITY_REPORT
tail -6 gpurun_out/r04/gpu_tests.log | cut -c1-300
bash scripts/profile_r04.sh 2>&1 | tail -60
