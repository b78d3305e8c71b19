#!/bin/bash
# round 6, run 8: k_matnet_bwd build variants (2 waves per SIMD forced / adjoints interleaved with the outer products) under the kernel trace of the training step
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
for v in base bw2 bwi bwi2 base bw2 bwi bwi2; do
  if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
  rm -rf gpurun_out/pf; mkdir -p gpurun_out/pf
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o t -- python3 scripts/train_step_bench.py --steps 3 > gpurun_out/pf/log_tr 2>&1
  f=$(find gpurun_out/pf/kt -name "*kernel_stats.csv" | head -1)
  echo "$v  $(grep '^stage-1' gpurun_out/pf/log_tr | cut -c1-45)  $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_matnet_bwd' in r['Name'] or 'k_direct_bwd' in r['Name']: print(r['Name'][4:16], '%.1f us' % (float(r['AverageNs'])/1e3), end='  ')
")"
done 2>&1 | tee gpurun_out/r06/ab_matnet_bwd_builds.txt
rm -rf gpurun_out/pf
