#!/bin/bash
# round 6, run 37: workgroups of the persistent traversal kernels per CU (MIRRES_TRACE_BLOCKS_PER_CU, in units of 256 threads: 8 = 32 waves per CU since round 4, when the
# shadow-ray kernel held 75-79 registers = six resident waves per SIMD; it holds 70 = seven now): 6 / 7 / 8 / 9 / 10 / 12 on 128-spp frames, interleaved, two rounds
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_trace_blocks.txt
one() { python3 bench.py --mesh $1 --no-extras --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
{ echo "# MIRRES_TRACE_BLOCKS_PER_CU; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do echo "-- $mesh"; for i in 1 2; do for b in 8 6 7 9 10 12; do echo "blocks $b  $(MIRRES_TRACE_BLOCKS_PER_CU=$b one $mesh)"; done; done; done
} 2>&1 | tee $O
