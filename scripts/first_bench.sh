#!/bin/bash
# first GPU pass: gpu tests, smoke, small bench, full bench, rocprof kernel trace
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
python bench.py --res 200 --ssaa 1 --spp 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_small.log 2>&1
timeout 900 python bench.py --steps 1 --warmup 1 > gpurun_out/bench_full.log 2>&1
