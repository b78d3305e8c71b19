#!/bin/bash
# Hardware counters of the backward kernels of the stage-1 training step (scripts/train_step_bench.py: 800 x 800, 32 spp), averages per launch; separate rocprofv3 --pmc
# passes (counters only; a set rocprofv3 rejects fails that pass alone). Output: gpurun_out/pmc_train/summary.json + a table.   usage: scripts/pmc_train.sh [tag]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
tag=${1:-train}
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 scripts/train_step_bench.py --steps 2 > $out/log$i.txt 2>&1 || echo "pass $i ($set) failed rc=$?"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT
GRBM_GUI_ACTIVE TA_TA_BUSY_sum
FETCH_SIZE
WRITE_SIZE
TCC_ATOMIC_sum TCC_REQ_sum
TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
SETS
python3 - "$out" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
regs = {}
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mr::', '')[:40]
        if 'bwd' not in k: continue
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
        regs[k] = (r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('LDS_Block_Size'), r.get('Scratch_Size'))
res = {}
for k, cs in agg.items():
    c = {n: x[0] / x[1] for n, x in cs.items()}; c['_launches'] = max(x[1] for x in cs.values())
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8.0
    if cyc and 'SQ_ACTIVE_INST_VALU' in c:
        c['kernel_cycles'] = cyc; c['launch_us_at_2.4GHz'] = cyc / 2400.0
        c['valu_busy'] = 4.0 * c['SQ_ACTIVE_INST_VALU'] / (1024.0 * cyc)
        c['lane_util'] = c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU']) if c['SQ_ACTIVE_INST_VALU'] else 0
        c['wait_any_of_wave_cycles'] = c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'] if c.get('SQ_WAVE_CYCLES') else 0
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c: c['hbm_bytes_per_launch_fetch2x_plus_write'] = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0
    c['_regs(vgpr,agpr,lds,scratch)'] = regs.get(k)
    res[k] = c
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
for k, c in sorted(res.items()):
    print(k)
    for n in sorted(c): print('   %-46s %s' % (n, ('%16.3f' % c[n]) if isinstance(c[n], float) else c[n]))
PY
rm -rf $out/p[0-9]*
