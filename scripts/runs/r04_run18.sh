#!/bin/bash
# Round 4, GPU run 18: experiment — spatial shadow rays whose answer the origin pixel already knows (it holds the same light sample, vcode set): MIRRES_SKIP_DEAD=2
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_known_rays.txt; : > $out
for mesh in icosphere clustered; do for f in 0 1 2; do echo "hash $mesh level $f: $(MIRRES_MESH=$mesh MIRRES_SKIP_DEAD=$f python3 scripts/dev_frame_hash.py 24 2>/dev/null | tail -1)" >> $out; done; done
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; shadow launch', r['launch_ms'], 'ms', r['grays_per_s'], 'Grays/s; not traced', r['rays_not_traced_frac'], 'known', r['rays_known_frac'])"; }
for rep in 1 2; do for mesh in icosphere clustered; do for f in 1 2; do echo "mesh $mesh level $f: $(MIRRES_SKIP_DEAD=$f one $mesh)" >> $out; done; done; done
cat $out
