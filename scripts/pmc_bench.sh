#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
rm -rf gpurun_out/pb1 gpurun_out/pb2; mkdir -p gpurun_out/pb1 gpurun_out/pb2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pb1 -o p -- python3 bench.py --spp 4 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/pb1/log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pb2 -o p -- python3 bench.py --spp 4 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/pb2/log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ('pb1','pb2'):
    fs = glob.glob('gpurun_out/%s/**/*counter_collection.csv' % d, recursive=True)
    if not fs: print(d, 'no csv'); print(open('gpurun_out/%s/log'%d).read()[-500:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0].replace('void ','').replace('mr::','')[:28]
        a = agg[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, cs in agg.items():
        if not k.startswith('k_'): continue
        v = {c: x[0]/x[1] for c, x in cs.items()}
        if d == 'pb1':
            wc = v.get('SQ_WAVE_CYCLES',1)
            print("%-28s n=%3d waveMcyc %8.1f wait_any %4.0f%% wait_inst %4.0f%% valu_util %4.0f%% valu_inst/M %7.2f vmem_rd/M %6.2f" % (k, list(cs.values())[0][1], wc/1e6, 100*v.get('SQ_WAIT_ANY',0)/wc, 100*v.get('SQ_WAIT_INST_ANY',0)/wc,
                  100*v.get('SQ_THREAD_CYCLES_VALU',0)/max(1,64*v.get('SQ_ACTIVE_INST_VALU',1)), v.get('SQ_INSTS_VALU',0)/1e6, v.get('SQ_INSTS_VMEM_RD',0)/1e6))
        else:
            print("%-28s L1acc/M %8.2f L1->L2/M %7.2f L1hit %4.0f%% L2hit %4.0f%% salu/M %7.2f waves %8.0f" % (k, v.get('TCP_TOTAL_CACHE_ACCESSES_sum',0)/1e6, v.get('TCP_TCC_READ_REQ_sum',0)/1e6,
                  100*(1-v.get('TCP_TCC_READ_REQ_sum',0)/max(1,v.get('TCP_TOTAL_CACHE_ACCESSES_sum',1))), 100*v.get('TCC_HIT_sum',0)/max(1,v.get('TCC_HIT_sum',0)+v.get('TCC_MISS_sum',0)), v.get('SQ_INSTS_SALU',0)/1e6, v.get('SQ_WAVES',0)))
PY
rm -rf gpurun_out/pb1 gpurun_out/pb2
