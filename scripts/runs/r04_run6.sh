#!/bin/bash
# Round 4, GPU run 6: reference bilateral kernels as a checker; bucket sort in front of the material lookup (A/B by MIRRES_GRID_SORT) + the frame tests that go through it
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time timeout 1500 python3 -m pytest tests/test_gpu_bilateral.py tests/test_gpu_matnet.py tests/test_gpu_fullsize.py tests/test_gpu_clustered.py -m gpu -q ) > gpurun_out/r04/gpu_tests_sort.log 2>&1
tail -8 gpurun_out/r04/gpu_tests_sort.log | cut -c1-300
out=gpurun_out/r04/ab_grid_sort.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do for f in 0 1; do echo "mesh $mesh grid_sort $f: $(MIRRES_GRID_SORT=$f one $mesh)" >> $out; done; done; done
cat $out
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for f in 0 1; do
  rm -rf gpurun_out/ks$f; MIRRES_GRID_SORT=$f timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks$f -o k -- python3 bench.py --spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/ks$f.log 2>&1
  echo "== MIRRES_GRID_SORT=$f"; find gpurun_out/ks$f -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
for r in csv.DictReader(open('{}')):
    n=r['Name']
    if any(k in n for k in ('k_mlp_mfma','k_active_from_live','k_ls_','k_bounce_gen','k_trace_closest4')): print('%-60s calls %4s avg %9.1f us total %8.2f ms' % (n.replace('void mr::','')[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
"
  rm -rf gpurun_out/ks$f
done > gpurun_out/r04/grid_sort_kernels.txt 2>&1
cat gpurun_out/r04/grid_sort_kernels.txt
