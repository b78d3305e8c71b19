#!/bin/bash
# VALU issue share of every kernel in the frame loop: SQ_INSTS_VALU summed per kernel over one serialised (MIRRES_STREAMS=1) 16-spp frame
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
export MIRRES_STREAMS=1
rm -rf gpurun_out/pv; mkdir -p gpurun_out/pv gpurun_out/out
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pv -o p -- python3 bench.py --spp 16 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/pv/log 2>&1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/pv/**/*counter_collection.csv', recursive=True)
if not fs: print(open('gpurun_out/pv/log').read()[-1500:])
else:
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ','').replace('mr::','')[:30]
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    tot = sum(cs['SQ_INSTS_VALU'][0] for cs in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1]['SQ_INSTS_VALU'][0])
    with open('gpurun_out/out/pmc_frame_valu.txt', 'w') as f:
        for k, cs in rows[:24]:
            line = "%-30s n %4d VALU %8.1fM (%4.1f%%) wavecyc %9.1fM busy %8.1fM vmem_rd %7.1fM salu %7.1fM wait %9.1fM" % (k, cs['SQ_INSTS_VALU'][1], cs['SQ_INSTS_VALU'][0]/1e6, 100*cs['SQ_INSTS_VALU'][0]/tot,
                   cs['SQ_WAVE_CYCLES'][0]/1e6, cs['SQ_BUSY_CYCLES'][0]/1e6, cs['SQ_INSTS_VMEM_RD'][0]/1e6, cs['SQ_INSTS_SALU'][0]/1e6, cs['SQ_WAIT_INST_ANY'][0]/1e6)
            print(line); f.write(line + "\n")
PY
rm -rf gpurun_out/pv
