#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc1 -o p -- python3 scripts/dev_trace_bench.py > gpurun_out/pmc1/log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc2 -o p -- python3 scripts/dev_trace_bench.py > gpurun_out/pmc2/log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc3 -o p -- python3 scripts/dev_trace_bench.py > gpurun_out/pmc3/log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc1','pmc2','pmc3'):
    fs = glob.glob('gpurun_out/%s/**/*counter_collection.csv' % d, recursive=True)
    if not fs: print(d, 'no csv'); print(open('gpurun_out/%s/log'%d).read()[-800:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'k_trace' not in k: continue
        k = 'any' if 'k_trace_any' in k else 'closest'
        k += '<cnt>' if 'ILb1' in r['Kernel_Name'] or '<true>' in r['Kernel_Name'] else ''
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, cs in agg.items():
        print(d, k, {c: round(v[0]/v[1]) for c, v in cs.items()})
PY
rm -rf gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
