#!/bin/bash
# Round 4, GPU run 31: block sizes / LDS stack of the other kernels re-checked on the final state (build variants in ab/): k_initial_gen 128 / 256 / 512 threads, k_bounce_gen 256 / 512,
# LDS part of the ordered closest-hit kernel's stack 8 / 12 / 16 entries
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_misc_knobs.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; closest', r['closest']['launch_ms'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do
  echo "mesh $mesh default: $(one $mesh)" >> $out
  for v in igen128 igen512 bgen256 ldsst8 ldsst16; do echo "mesh $mesh $v: $(MIRRES_LIB=$PWD/ab/libmirres_$v.so one $mesh)" >> $out; done
done; done
cat $out
