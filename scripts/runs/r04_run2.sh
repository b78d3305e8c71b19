#!/bin/bash
# Round 4, GPU run 2: the whole -m gpu suite (no -x), default bench line
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
export MIRRES_PARITY_REPORT=$PWD/gpurun_out/r04/parity_report.txt; rm -f $MIRRES_PARITY_REPORT gpurun_out/clustered_mesh_report.txt
( time timeout 2400 python3 -m pytest tests -m gpu -q ) > gpurun_out/r04/gpu_tests.log 2>&1
unset MIRRES_PARITY_REPORT
tail -15 gpurun_out/r04/gpu_tests.log
( time python3 bench.py ) > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
tail -3 gpurun_out/r04/bench_default.err
cat gpurun_out/clustered_mesh_report.txt gpurun_out/deep_tree_stack.txt
