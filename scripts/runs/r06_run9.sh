#!/bin/bash
# round 6, run 9: RCCL tests incl. the native halo exchange; host / device cost of the exchange, callback vs native; k_matnet_bwd build variants
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -12 | tee gpurun_out/r06/tests_run9.txt
for mesh in icosphere clustered; do MIRRES_MESH=$mesh timeout 600 python3 scripts/dev_halo_host_cost.py 512 2>&1 | grep -v "^\[W\|amdgpu.ids\|version\|Hostname\|Librccl" | tail -10; done | tee gpurun_out/r06/halo_host_cost.txt
for v in base bw2 bwi bwi2 base bw2 bwi bwi2; do
  if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
  rm -rf gpurun_out/pf; mkdir -p gpurun_out/pf
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o t -- python3 scripts/train_step_bench.py --steps 3 > gpurun_out/pf/log_tr 2>&1 || tail -3 gpurun_out/pf/log_tr
  f=$(find gpurun_out/pf/kt -name "*kernel_stats.csv" | head -1)
  echo "$v  $(grep '^stage-1' gpurun_out/pf/log_tr | cut -c1-45)  $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_matnet_bwd' in r['Name'] or 'k_direct_bwd' in r['Name']: print(r['Name'][4:16], '%.1f us' % (float(r['AverageNs'])/1e3), end='  ')
")"
done 2>&1 | tee gpurun_out/r06/ab_matnet_bwd_builds.txt
unset MIRRES_LIB
rm -rf gpurun_out/pf
