#!/bin/bash
# round 6, run 18: two-step hierarchy build (SAH top only for long frames): the whole GPU suite, the training step, the metric's frame
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r06/gpu_suite_b.txt
for i in 1 2 3; do timeout 300 python3 scripts/train_step_bench.py --steps 6 2>&1 | grep '^stage-1' | cut -c1-60; done | tee gpurun_out/r06/train_step_c.txt
for mesh in icosphere clustered; do timeout 300 python3 bench.py --mesh $mesh --no-extras --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mesh', d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; done
