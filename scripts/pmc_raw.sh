#!/bin/bash
# raw SQ counters of the traversal kernels in the frame loop (one pass), printed per kernel as averages per launch
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
rm -rf gpurun_out/pr; mkdir -p gpurun_out/pr
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pr -o p -- python3 bench.py --spp 4 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/pr/log 2>&1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/pr/**/*counter_collection.csv', recursive=True)
if not fs: print(open('gpurun_out/pr/log').read()[-1500:])
else:
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ','').replace('mr::','')[:28]
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, cs in agg.items():
        if not k.startswith('k_trace') and not k.startswith('k_initial_gen'): continue
        print(k, {c: round(x[0]/x[1]/1e6, 3) for c, x in cs.items()})
PY
rm -rf gpurun_out/pr
