#!/bin/bash
# Round 4, GPU run 32: 16-byte light-tile records in the batched initial resampling (MIRRES_TILE_COMPACT=0: the 32-byte records): frame parity tests, A/B with the kernel's own time
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time timeout 1500 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_clustered.py tests/test_gpu_fullsize.py -m gpu -q ) 2>&1 | tail -5 | cut -c1-300
out=gpurun_out/r04/ab_tile_compact.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do for mesh in icosphere clustered; do for f in 0 1; do echo "mesh $mesh tile_compact $f: $(MIRRES_TILE_COMPACT=$f one $mesh)" >> $out; done; done; done
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for f in 0 1; do
  rm -rf gpurun_out/ks$f; MIRRES_TILE_COMPACT=$f timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks$f -o k -- python3 bench.py --spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/ks$f.log 2>&1
  echo "== MIRRES_TILE_COMPACT=$f" >> $out; find gpurun_out/ks$f -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv
for r in csv.DictReader(open('{}')):
    n=r['Name']
    if any(k in n for k in ('k_initial_gen','k_tile_aux')): print('%-60s calls %4s avg %9.1f us total %8.2f ms' % (n.replace('void mr::','').replace('mr::','')[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
" >> $out
  rm -rf gpurun_out/ks$f
done
cat $out
