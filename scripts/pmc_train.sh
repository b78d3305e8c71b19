#!/bin/bash
# SQ counters of the backward kernels of the stage-1 training step (scripts/train_step_bench.py), averages per launch
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
rm -rf gpurun_out/pt; mkdir -p gpurun_out/pt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d gpurun_out/pt -o p -- python3 scripts/train_step_bench.py --steps 2 > gpurun_out/pt/log 2>&1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/pt/**/*counter_collection.csv', recursive=True)
if not fs: print(open('gpurun_out/pt/log').read()[-1500:])
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
for fn in fs:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0].replace('void ','').replace('mr::','')[:28]
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, cs in agg.items():
    if 'bwd' not in k: continue
    print(k, {c: round(x[0]/x[1]/1e6, 3) for c, x in cs.items()}, 'launches', list(cs.values())[0][1])
PY
rm -rf gpurun_out/pt
