#!/bin/bash
# Round 4, GPU run 26: the LDS-staged first four levels in the ordered closest-hit kernel (MIRRES_CLOSEST_TOP=0 reads them from global memory as before): trace parity tests, A/B
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time timeout 1500 python3 -m pytest tests/test_gpu_bvh.py tests/test_gpu_clustered.py tests/test_gpu_render.py -m gpu -q ) 2>&1 | tail -5 | cut -c1-300
out=gpurun_out/r04/ab_closest_top.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=r['closest']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; closest launch', c['launch_ms'], 'ms', c['grays_per_s'], 'Grays/s', c['bytes_per_ray'], 'B/ray from global memory')"; }
for rep in 1 2 3; do for mesh in icosphere clustered; do for f in 0 1; do echo "mesh $mesh closest_top $f: $(MIRRES_CLOSEST_TOP=$f one $mesh)" >> $out; done; done; done
for mesh in icosphere clustered; do for f in 0 1; do echo "hash $mesh closest_top $f: $(MIRRES_MESH=$mesh MIRRES_CLOSEST_TOP=$f python3 scripts/dev_frame_hash.py 12 2>/dev/null | tail -1)" >> $out; done; done
cat $out
