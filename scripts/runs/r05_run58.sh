#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# samples per batch (MIRRES_PT_BATCH; default 32) against the frame's sample count: ms per frame (bench.py --no-extras --no-cpu-baseline --no-roofline, 3 + 1 frames) and the stage-1 training step; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for spp in 32 64 128; do for k in 4 8 16 32; do
    echo "spp $spp, batch $k: $(MIRRES_PT_BATCH=$k timeout 300 python3 bench.py --spp $spp --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms", d["value"], "Msamples/s")')"
  done; done
  for k in 4 8 16 32; do echo "training step, batch $k: $(MIRRES_PT_BATCH=$k timeout 300 python3 scripts/train_step_bench.py --steps 5 2>&1 | grep '^stage-1' | cut -c1-70)"; done
} 2>&1 | tee gpurun_out/r05/batch_vs_spp.txt
