#!/bin/bash
# Round 4, GPU run 1: the whole -m gpu suite (new clustered / many-sample / deep-tree tests), the default bench line (both meshes + train step), stack-size A/B
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
export MIRRES_PARITY_REPORT=$PWD/gpurun_out/r04/parity_report.txt; rm -f $MIRRES_PARITY_REPORT gpurun_out/clustered_mesh_report.txt
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r04/gpu_tests.log 2>&1
unset MIRRES_PARITY_REPORT
tail -5 gpurun_out/r04/gpu_tests.log
( time python3 bench.py ) > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
tail -3 gpurun_out/r04/bench_default.err
bash scripts/dev_ab.sh 0 stack64 > gpurun_out/r04/ab_stack_micro.txt 2>&1
bash scripts/dev_ab_frame.sh stack64 > gpurun_out/r04/ab_stack_frame.txt 2>&1
cat gpurun_out/r04/ab_stack_micro.txt gpurun_out/r04/ab_stack_frame.txt
