#!/bin/bash
# Round 4, GPU run 28: default without the LDS-staged top levels in the shadow-ray kernel: traversal / frame parity tests, then refill threshold re-checked on the new default (build variants)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time timeout 1500 python3 -m pytest tests/test_gpu_bvh.py tests/test_gpu_clustered.py tests/test_gpu_render.py tests/test_gpu_layout.py tests/test_gpu_dump.py -m gpu -q ) 2>&1 | tail -5 | cut -c1-300
out=gpurun_out/r04/ab_refill_notop.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; shadow launch', r['launch_ms'], 'ms', r['grays_per_s'], 'Grays/s; closest', r['closest']['launch_ms'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do
  echo "mesh $mesh refill 40 (default): $(one $mesh)" >> $out
  for r in 32 48; do echo "mesh $mesh refill $r: $(MIRRES_LIB=$PWD/ab/libmirres_refill$r.so one $mesh)" >> $out; done
done; done
cat $out
