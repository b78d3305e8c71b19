#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout -k 5 1800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05/gputests_run16.txt 2>&1; tail -4 gpurun_out/r05/gputests_run16.txt
python3 bench.py --steps 3 --warmup 1 > gpurun_out/r05/bench_run16.json 2> gpurun_out/r05/bench_run16.err; tail -c 600 gpurun_out/r05/bench_run16.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_run16.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('valu_useful'), d['roofline']['launch_ms'], d['cpu_baseline'])
c=d.get('clustered',{}); print('clustered', c.get('value'), c.get('ms_per_step'), c.get('steps'), (c.get('roofline') or {}).get('launch_ms'))
print('train', d.get('train_step'))
PY
MIRRES_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --spp 16 --steps 2 --warmup 2 --no-roofline > gpurun_out/r05/bench_two_ranks_gloo.json 2> gpurun_out/r05/bench_two_ranks_gloo.err; tail -c 1500 gpurun_out/r05/bench_two_ranks_gloo.json | cut -c1-1500
