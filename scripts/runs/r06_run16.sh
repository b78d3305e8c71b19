#!/bin/bash
# round 6, run 16: samples per batch for the 32-spp training frame (one batch of 32 = the initial stage before, the final stage after the whole chain)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
{ echo "# training step (800 x 800, 32 spp) by MIRRES_PT_BATCH; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
for k in 32 16 8 11 32 16 8 11; do echo "batch $k: $(MIRRES_PT_BATCH=$k timeout 300 python3 scripts/train_step_bench.py --steps 6 2>&1 | grep '^stage-1' | cut -c29-60)"; done; } | tee gpurun_out/r06/ab_train_batch.txt
