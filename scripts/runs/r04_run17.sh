#!/bin/bash
# Round 4, GPU run 17 (third pass): luminance rule with lum in the first half of the packed reservoir record (no extra load) against tracing every spatial shadow ray (0)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time timeout 2000 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_clustered.py tests/test_gpu_fullsize.py -m gpu -q ) > gpurun_out/r04/gpu_tests_skipdead.log 2>&1
tail -8 gpurun_out/r04/gpu_tests_skipdead.log | cut -c1-400
out=gpurun_out/r04/ab_skip_dead.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; shadow launch', r['launch_ms'], 'ms', r['grays_per_s'], 'Grays/s; not traced', r['rays_not_traced_frac'], 'per_ray', r['per_ray']['any_production'])"; }
for rep in 1 2 3; do for mesh in icosphere clustered; do for f in 0 1; do echo "mesh $mesh skip_dead $f: $(MIRRES_SKIP_DEAD=$f one $mesh)" >> $out; done; done; done
cat $out
