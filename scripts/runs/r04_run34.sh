#!/bin/bash
# Round 4, GPU run 34: spatial-pass launch shapes re-checked on the final state (build variants): k_spatial_resolve forced to four waves per SIMD, 8 x 8 resolve tiles, 64 px XCD chunks
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_spatial_shapes.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do
  echo "mesh $mesh default: $(one $mesh)" >> $out
  for v in sres4 srest8 chunk64; do echo "mesh $mesh $v: $(MIRRES_LIB=$PWD/ab/libmirres_$v.so one $mesh)" >> $out; done
done; done
cat $out
