#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# samples per batch 32 (default) against 64 at 512 spp: bench.py --no-extras --no-cpu-baseline --no-roofline, 3 + 1 frames; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do for k in 32 64 32 64; do
    echo "$mesh, batch $k: $(MIRRES_PT_BATCH=$k timeout 300 python3 bench.py --mesh $mesh --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms", d["value"], "Msamples/s")')"
  done; done; } 2>&1 | tee gpurun_out/r05/batch64.txt
