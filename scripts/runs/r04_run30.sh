#!/bin/bash
# Round 4, GPU run 30: workgroups per CU launched by the persistent traversal kernels (MIRRES_TRACE_BLOCKS_PER_CU; six fit at once), finer sweep, 128 spp and the metric's 512 spp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_trace_blocks.txt; : > $out
one() { python3 bench.py --mesh $1 --spp $2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; shadow launch', r['launch_ms'], 'ms; closest', r['closest']['launch_ms'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do for b in 6 7 8 10 12 16; do echo "mesh $mesh 128 spp blocks per CU $b: $(MIRRES_TRACE_BLOCKS_PER_CU=$b one $mesh 128)" >> $out; done; done; done
for mesh in icosphere clustered; do for b in 6 8 12; do echo "mesh $mesh 512 spp blocks per CU $b: $(MIRRES_TRACE_BLOCKS_PER_CU=$b one $mesh 512)" >> $out; done; done
cat $out
