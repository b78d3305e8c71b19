#!/bin/bash
# round 6, run 17: the training step by hierarchy builder (MIRRES_PRIVATE_TREE: 2 = extended Morton + SAH top (default), 1 = extended Morton only, 0 = collapsed reference LBVH)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
{ echo "# training step (800 x 800, 32 spp, LBVH rebuild per step) by MIRRES_PRIVATE_TREE; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
for k in 2 1 0 2 1 0; do echo "private tree $k: $(MIRRES_PRIVATE_TREE=$k timeout 300 python3 scripts/train_step_bench.py --steps 6 2>&1 | grep '^stage-1' | cut -c29-60)"; done; } | tee gpurun_out/r06/ab_train_tree.txt
