#!/bin/bash
# Round 4, GPU run 36: workgroup size of the persistent traversal kernels (no workgroup-level cooperation is left in them): 64 / 128 / 256 threads, the same number of waves launched
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_trace_block.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; shadow launch', r['launch_ms'], 'ms; closest', r['closest']['launch_ms'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do for v in tb256 tb128 tb64; do echo "mesh $mesh $v: $(MIRRES_LIB=$PWD/ab/libmirres_$v.so one $mesh)" >> $out; done; done; done
for mesh in icosphere clustered; do echo "hash $mesh tb64: $(MIRRES_MESH=$mesh MIRRES_LIB=$PWD/ab/libmirres_tb64.so python3 scripts/dev_frame_hash.py 12 2>/dev/null | tail -1)" >> $out; done
cat $out
