#!/bin/bash
# Round 4, GPU run 21: timing experiment — the cost of the refill's two dependent gathers in the pixel-pair mode of the shadow-ray kernel (a build that issues a second pair of them)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_refill_gathers.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; shadow launch', r['launch_ms'], 'ms', r['grays_per_s'], 'Grays/s')"; }
for rep in 1 2; do for mesh in icosphere clustered; do
  echo "mesh $mesh production:        $(one $mesh)" >> $out
  echo "mesh $mesh refill gathers x2: $(MIRRES_LIB=$PWD/ab/libmirres_refill2x.so one $mesh)" >> $out
done; done
cat $out
