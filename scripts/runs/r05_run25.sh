#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout -k 5 1800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05/gputests_run25.txt 2>&1; tail -4 gpurun_out/r05/gputests_run25.txt
python3 bench.py --steps 4 --warmup 2 > gpurun_out/r05/bench_run25.json 2> gpurun_out/r05/bench_run25.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_run25.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['closest']['launch_ms'], d['cpu_baseline']['max_abs_err'])
c=d.get('clustered',{}); print('clustered', c.get('value'), c.get('ms_per_step'), (c.get('roofline') or {}).get('launch_ms'), (c.get('roofline') or {}).get('closest',{}).get('launch_ms'))
print('train', d.get('train_step',{}).get('ms_per_step'))
PY
