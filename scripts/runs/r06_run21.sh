#!/bin/bash
# round 6, run 21: the N > 1 path of bench.py as a two-rank gloo dry run on one GPU, and its watchdog (a hang of the second scheme must not take the first scheme's line with it)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
MIRRES_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29741 bench.py --gpus 2 --steps 2 --warmup 2 --spp 32 --no-roofline 2>/dev/null | tail -1 > gpurun_out/r06/r06_bench_two_ranks_gloo_dry_run.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r06/r06_bench_two_ranks_gloo_dry_run.json').read()); print('normal: value', d['value'], d['config'].get('value_scheme'), 'strips', d.get('strips', {}).get('value'), d['config'].get('strip_balance'))"
MIRRES_BENCH_WATCHDOG_S=0.3 MIRRES_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29742 bench.py --gpus 2 --steps 2 --warmup 2 --spp 32 --no-roofline 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r06/r06_bench_watchdog_line.json; echo "watchdog run exit $?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r06/r06_bench_watchdog_line.json').read()); print('watchdog: value', d['value'], d['config'].get('value_scheme'), 'strips', d.get('strips'))"
