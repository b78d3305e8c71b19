#!/bin/bash
# round 6, run 30: why the leaf queue loses — the counting kernel's leaf statistics and two counter passes (VALU / scalar) for the base kernel and the lq32_16 variant
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06 gpurun_out/lqpmc
O=gpurun_out/r06/ab_leaf_queue_counters.txt
{ echo "# leaf queue (lq32_16) against the shipped kernel: counting-kernel statistics on the frame's own rays, then hardware counters of the 6.9 M / 9.0 M-ray microbenchmark"
  for v in base lq32_16; do
    if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
    echo "## $v"; timeout 300 python3 scripts/dev_leaf_branch.py 4 2>&1 | grep -v amdgpu.ids | tail -2
    for mesh in icosphere clustered; do
      export MIRRES_MESH=$mesh; i=0
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_WAVES"; do
        i=$((i+1)); d=gpurun_out/lqpmc/${v}_${mesh}_$i; rm -rf $d
        timeout -k 5 100 rocprofv3 --pmc $set --output-format csv -d $d -o p -- python3 scripts/dev_any_pmc.py 1600 7 3 0 > $d.log 2>&1 || echo "pass failed rc=$?"
      done
      python3 - $v $mesh <<'PY'
import csv, glob, collections, sys
v, mesh = sys.argv[1:3]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/lqpmc/%s_%s_*/**/*counter_collection.csv' % (v, mesh), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_trace_any4q' not in r['Kernel_Name']: continue
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
c = {k: x[0] / x[1] for k, x in agg.items()}
print('%s %s: ' % (v, mesh) + ', '.join('%s %.4g' % (k, c[k]) for k in sorted(c)))
if 'SQ_THREAD_CYCLES_VALU' in c and 'SQ_ACTIVE_INST_VALU' in c:
    print('   lane utilisation %.3f' % (c['SQ_THREAD_CYCLES_VALU'] / (64.0 * c['SQ_ACTIVE_INST_VALU'])))
PY
    done; unset MIRRES_MESH
  done
} 2>&1 | tee $O
