#!/bin/bash
# round 6, run 39: samples per batch 64 / 48 against 32 at 512 spp, interleaved, three rounds, both meshes (run 38 showed +0 ... +1.5 % within its noise)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_pt_batch.txt
one() { python3 bench.py --mesh $1 --no-extras --spp 512 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
{ echo "# MIRRES_PT_BATCH at 512 spp; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do echo "-- $mesh"; for i in 1 2 3; do for k in 32 64 48; do echo "batch $k  $(MIRRES_PT_BATCH=$k one $mesh)"; done; done; done
} 2>&1 | tee $O
