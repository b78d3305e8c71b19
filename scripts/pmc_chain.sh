#!/bin/bash
# PMC passes over ONE serialised frame of the bench (MIRRES_STREAMS=1, 8 spp): per-kernel SQ / TA / TCP counters of EVERY kernel of the per-sample
# chain (k_spatial_gen, k_spatial_resolve, k_initial_gen, ...), per-launch averages, on one mesh.  Separate passes, counters only.
# usage: scripts/pmc_chain.sh <tag> <mesh> [spp]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
tag=${1:-chain}; mesh=${2:-icosphere}; spp=${3:-8}
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
export MIRRES_STREAMS=1
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p -- python3 bench.py --mesh $mesh --spp $spp --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras > $out/log$i.txt 2>&1 || echo "pass $i ($set) failed rc=$?"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES
GRBM_GUI_ACTIVE TA_TA_BUSY_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM
SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32
FETCH_SIZE
WRITE_SIZE
SETS
python3 - "$out" "$mesh" "$spp" <<'PY'
import csv, glob, collections, sys, json
out, mesh, spp = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mr::', '')[:48]
        if not k.startswith('k_'): continue
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
        agg[k]['_vgpr'] = [float(r.get('VGPR_Count') or r.get('Arch_VGPR_Count') or 0), 1]
        agg[k]['_lds'] = [float(r.get('LDS_Block_Size') or 0), 1]
        agg[k]['_scratch'] = [float(r.get('Scratch_Size') or 0), 1]
res = {}
for k, cs in agg.items():
    d = {c: x[0] / max(1, x[1]) for c, x in cs.items()}
    n = max((x[1] for c, x in cs.items() if not c.startswith('_')), default=0)
    d['_launches'] = n
    g = d.get('GRBM_GUI_ACTIVE')
    if g and d.get('SQ_ACTIVE_INST_VALU') is not None:
        # as profiles/README.md (round 3): SQ_ACTIVE_INST_VALU is summed over the SIMDs in quad-cycles; GRBM_GUI_ACTIVE summed over the 8 XCDs
        d['valu_busy'] = 4.0 * d['SQ_ACTIVE_INST_VALU'] / (1024.0 * g / 8.0)
        d['kernel_cycles'] = g / 8.0
    if d.get('SQ_ACTIVE_INST_VALU'):
        d['lane_util'] = d.get('SQ_THREAD_CYCLES_VALU', 0) / (64.0 * d['SQ_ACTIVE_INST_VALU'])
    if d.get('SQ_WAVE_CYCLES'):
        d['wait_any_of_wave_cycles'] = d.get('SQ_WAIT_ANY', 0) / d['SQ_WAVE_CYCLES']
        d['wait_inst_of_wave_cycles'] = d.get('SQ_WAIT_INST_ANY', 0) / d['SQ_WAVE_CYCLES']
    if d.get('TCP_TOTAL_CACHE_ACCESSES_sum'):
        d['l1_hit'] = 1.0 - d.get('TCP_TCC_READ_REQ_sum', 0) / d['TCP_TOTAL_CACHE_ACCESSES_sum']
    if g:
        cyc = g / 8.0
        if 'TA_TA_BUSY_sum' in d: d['ta_busy'] = d['TA_TA_BUSY_sum'] / (256.0 * cyc)
        if 'TCP_PENDING_STALL_CYCLES_sum' in d: d['tcp_pending_stall'] = d['TCP_PENDING_STALL_CYCLES_sum'] / (256.0 * cyc)
        if 'TCP_TCR_TCP_STALL_CYCLES_sum' in d: d['tcp_tcr_stall'] = d['TCP_TCR_TCP_STALL_CYCLES_sum'] / (256.0 * cyc)
        if 'TA_ADDR_STALLED_BY_TC_CYCLES_sum' in d: d['ta_addr_stalled_by_tc'] = d['TA_ADDR_STALLED_BY_TC_CYCLES_sum'] / (256.0 * cyc)
    if d.get('TCP_TCC_READ_REQ_sum') and d.get('TCP_TCP_LATENCY_sum'):
        d['l1_miss_latency_cycles'] = d['TCP_TCP_LATENCY_sum'] / d['TCP_TCC_READ_REQ_sum']
    if d.get('SQ_WAVES') and d.get('SQ_INSTS_VALU') is not None:
        d['valu_insts_per_wave'] = d['SQ_INSTS_VALU'] / d['SQ_WAVES']
    if 'FETCH_SIZE' in d:
        d['hbm_bytes_per_launch_fetch2x_plus_write'] = (2 * d['FETCH_SIZE'] + d.get('WRITE_SIZE', 0)) * 1024
    res[k] = d
json.dump({'mesh': mesh, 'spp': spp, 'command': 'MIRRES_STREAMS=1 rocprofv3 --pmc <set> -- python3 bench.py --mesh %s --spp %d --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras' % (mesh, spp), 'kernels': res},
          open(out + '/summary.json', 'w'), indent=1)
keys = ['_launches', 'kernel_cycles', 'valu_busy', 'lane_util', 'wait_any_of_wave_cycles', 'l1_hit', 'ta_busy', 'tcp_pending_stall', 'tcp_tcr_stall', 'l1_miss_latency_cycles', 'valu_insts_per_wave', '_vgpr', 'hbm_bytes_per_launch_fetch2x_plus_write']
print('%-48s ' % 'kernel' + ' '.join('%12s' % k[:12] for k in keys))
for k in sorted(res, key=lambda k: -(res[k].get('kernel_cycles', 0) * res[k]['_launches'])):
    print('%-48s ' % k + ' '.join('%12.4g' % res[k].get(c, float('nan')) for c in keys))
PY
rm -rf $out/p[0-9]*
