#!/bin/bash
# Round 4, GPU run 37: waves launched per CU by the traversal kernels re-swept with single-wave workgroups (MIRRES_TRACE_BLOCKS_PER_CU x 4 waves; 8 = default), 512 spp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_trace_blocks_64.txt; : > $out
one() { python3 bench.py --mesh $1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do for b in 7 8 9 10; do echo "mesh $mesh 512 spp blocks per CU $b: $(MIRRES_TRACE_BLOCKS_PER_CU=$b one $mesh)" >> $out; done; done; done
cat $out
