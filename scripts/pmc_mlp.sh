#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 scripts/dev_mlp_bench.py 2>&1 | tail -2
rm -rf gpurun_out/pk; mkdir -p gpurun_out/pk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk -o k -- python3 scripts/dev_mlp_bench.py > gpurun_out/pk/log 2>&1
grep -E 'k_mlp_mfma|k_matnet_fwd' gpurun_out/pk/k_kernel_stats.csv | cut -c1-60,200-330
rm -rf gpurun_out/pk
rm -rf gpurun_out/pm; mkdir -p gpurun_out/pm
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d gpurun_out/pm -o p -- python3 scripts/dev_mlp_bench.py > gpurun_out/pm/log 2>&1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/pm/**/*counter_collection.csv', recursive=True)
if not fs: print(open('gpurun_out/pm/log').read()[-600:])
else:
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(fs[0])):
        if 'k_mlp_mfma' not in r['Kernel_Name']: continue
        a = agg['k_mlp_mfma'][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, cs in agg.items():
        v = {c: x[0] / x[1] for c, x in cs.items()}
        print(k, {c: round(x) for c, x in v.items()})
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v:
            print("MfmaUtil = MFMA_BUSY / (GUI_ACTIVE * 1024 SIMDs) = %.1f %%" % (100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] * 1024)))
PY
rm -rf gpurun_out/pm
