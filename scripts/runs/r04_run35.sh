#!/bin/bash
# Round 4, GPU run 35: 8 x 8 resolve tiles as the default: frame parity tests, then 8 x 8 generation tiles and 64 px chunks on top (build variants)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time timeout 1500 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_passes.py tests/test_gpu_fullsize.py -m gpu -q ) 2>&1 | tail -4 | cut -c1-300
out=gpurun_out/r04/ab_spatial_shapes2.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do
  echo "mesh $mesh default (8 x 8 resolve tiles): $(one $mesh)" >> $out
  for v in sgen8 sres8x; do echo "mesh $mesh $v: $(MIRRES_LIB=$PWD/ab/libmirres_$v.so one $mesh)" >> $out; done
done; done
cat $out
