#!/bin/bash
# Round 4, GPU run 27: the shadow-ray kernel with / without its LDS-staged top levels (MIRRES_TOPQ=0: every node from global memory), re-measured on the final kernel
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_any_top.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; shadow launch', r['launch_ms'], 'ms', r['grays_per_s'], 'Grays/s')"; }
for rep in 1 2 3; do for mesh in icosphere clustered; do for f in 85 0; do echo "mesh $mesh topq $f: $(MIRRES_TOPQ=$f one $mesh)" >> $out; done; done; done
cat $out
