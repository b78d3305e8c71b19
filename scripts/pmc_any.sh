#!/bin/bash
# PMC passes (separate runs, counters only, each under its own timeout: a counter set rocprofv3 rejects must not hang the box) over
# scripts/dev_any_pmc.py: per-kernel averages per launch for the traversal kernels
# usage: scripts/pmc_any.sh <tag> [res K launches mode]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
tag=${1:-any}; shift
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
python3 scripts/dev_any_pmc.py "$@" > $out/plain.txt 2>&1
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 scripts/dev_any_pmc.py "$@" > $out/log$i.txt 2>&1 || echo "pass $i ($set) failed rc=$?"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES
TA_TA_BUSY_sum GRBM_GUI_ACTIVE
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_TCP_LATENCY_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
SETS
python3 - "$out" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mr::', '')[:40]
        if not k.startswith('k_trace'): continue
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
res = {k: {c: x[0] / x[1] for c, x in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
print(open(out + '/plain.txt').read().strip().splitlines()[-1])
for k, cs in res.items():
    print(k)
    for c in sorted(cs): print('   %-42s %14.0f' % (c, cs[c]))
PY
rm -rf $out/p[0-9]*
