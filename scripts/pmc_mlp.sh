#!/bin/bash
# SQ counters of the MLP GEMM-phase kernel (scripts/dev_mlp_bench.py), averages per launch
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
rm -rf gpurun_out/pm; mkdir -p gpurun_out/pm gpurun_out/out
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/pm -o p -- python3 scripts/dev_mlp_bench.py > gpurun_out/pm/log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d gpurun_out/pm/b -o p -- python3 scripts/dev_mlp_bench.py > gpurun_out/pm/log2 2>&1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/pm/**/*counter_collection.csv', recursive=True)
if not fs: print(open('gpurun_out/pm/log').read()[-1500:])
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
for fn in fs:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0].replace('void ','').replace('mr::','')[:28]
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, cs in agg.items():
    if 'mlp_mfma' not in k: continue
    print(k, {c: round(x[0]/x[1]/1e6, 3) for c, x in cs.items()}, 'launches', list(cs.values())[0][1])
PY
tail -3 gpurun_out/pm/log | cut -c1-300
rm -rf gpurun_out/pm
