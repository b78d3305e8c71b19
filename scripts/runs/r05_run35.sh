#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/out; mkdir -p $O
timeout -k 5 600 python3 bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err
timeout -k 5 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_driver_style.json 2> $O/r05_bench_driver_style.err
timeout -k 5 600 python3 bench.py --mesh clustered --no-extras > $O/r05_bench_clustered.json 2> $O/r05_bench_clustered.err
python3 - <<'PY'
import json
for f in ("r05_bench_default","r05_bench_driver_style","r05_bench_clustered"):
    d=json.loads(open('gpurun_out/out/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['frac'], r.get('valu_useful'), r.get('frame_valu_issue',{}).get('frac'), (d.get('clustered') or {}).get('value'), ((d.get('clustered') or {}).get('roofline') or {}).get('frame_valu_issue',{}).get('frac'))
PY
