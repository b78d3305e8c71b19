#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# stream count (MIRRES_STREAMS; default 2) on the whole frame at 512 spp, 3 + 1 frames; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in clustered icosphere; do for st in 2 3 4 5 2 3; do
    echo "$mesh, streams $st: $(MIRRES_STREAMS=$st timeout 300 python3 bench.py --mesh $mesh --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms", d["value"], "Msamples/s")')"
  done; done; } 2>&1 | tee gpurun_out/r05/streams_512.txt
