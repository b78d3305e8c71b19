#!/bin/bash
# Round 4, GPU run 29: shadow-ray kernel knobs re-checked on the default without the LDS top: LDS part of the private stack (8 / 12 / 16 / 24 entries), resident workgroups per CU
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_any_lds_stack.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; shadow launch', r['launch_ms'], 'ms', r['grays_per_s'], 'Grays/s')"; }
for rep in 1 2; do for mesh in icosphere clustered; do
  echo "mesh $mesh lds 16 (default): $(one $mesh)" >> $out
  for r in 8 12 24; do echo "mesh $mesh lds $r: $(MIRRES_LIB=$PWD/ab/libmirres_lds$r.so one $mesh)" >> $out; done
  for b in 4 5 8; do echo "mesh $mesh lds 16 blocks per CU $b: $(MIRRES_TRACE_BLOCKS_PER_CU=$b one $mesh)" >> $out; done
done; done
cat $out
