#!/bin/bash
# Round 4, GPU run 33: candidate records of the batched initial resampling requested MR_IGEN_AHEAD at a time (1 / 4 / 8 = default / 16; build variants), tile-size modulo as a mask
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time timeout 900 python3 -m pytest tests/test_gpu_render.py -m gpu -q ) 2>&1 | tail -4 | cut -c1-300
out=gpurun_out/r04/ab_igen_ahead.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do
  echo "mesh $mesh ahead 8 (default): $(one $mesh)" >> $out
  for v in 1 4 16; do echo "mesh $mesh ahead $v: $(MIRRES_LIB=$PWD/ab/libmirres_ahead$v.so one $mesh)" >> $out; done
done; done
for mesh in icosphere clustered; do echo "hash $mesh ahead 8: $(MIRRES_MESH=$mesh python3 scripts/dev_frame_hash.py 12 2>/dev/null | tail -1)" >> $out; echo "hash $mesh 32-byte records: $(MIRRES_MESH=$mesh MIRRES_TILE_COMPACT=0 python3 scripts/dev_frame_hash.py 12 2>/dev/null | tail -1)" >> $out; done
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/ks; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o k -- python3 bench.py --spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/ks.log 2>&1
find gpurun_out/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv
for r in csv.DictReader(open('{}')):
    n=r['Name']
    if 'k_initial_gen' in n: print('%-60s calls %4s avg %9.1f us' % (n.replace('void mr::','').replace('mr::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
" >> $out
rm -rf gpurun_out/ks
cat $out
