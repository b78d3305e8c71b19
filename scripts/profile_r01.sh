#!/bin/bash
# round-1 profiles: kernel trace of the default bench command + PMC passes (HBM traffic) on a short run
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_kt/bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- python3 bench.py --spp 4 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/prof_fetch/bench.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o write -- python3 bench.py --spp 4 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/prof_write/bench.log 2>&1
tail -3 gpurun_out/prof_kt/bench.log; tail -3 gpurun_out/prof_fetch/bench.log; find gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write -type f | head -30; du -sh gpurun_out/*
python3 - <<'PY'
import csv, glob, collections, json
def load(pat):
    fs = glob.glob(pat, recursive=True)
    return fs[0] if fs else None
f = load('gpurun_out/prof_kt/**/*kernel_stats.csv')
if f:
    rows = list(csv.DictReader(open(f)))
    print('kernel_stats', f, len(rows))
    for r in rows[:14]:
        print(r)
out = {}
for name, pat in (('FETCH_SIZE', 'gpurun_out/prof_fetch/**/*counter_collection.csv'), ('WRITE_SIZE', 'gpurun_out/prof_write/**/*counter_collection.csv')):
    f = load(pat)
    if not f: print('no', name); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') != name: continue
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    out[name] = {k: {'sum': v[0], 'n': v[1], 'avg': v[0] / v[1]} for k, v in agg.items()}
    for k, v in sorted(out[name].items(), key=lambda kv: -kv[1]['sum'])[:12]:
        print(name, k, v)
json.dump(out, open('gpurun_out/pmc_raw.json', 'w'), indent=1)
PY
