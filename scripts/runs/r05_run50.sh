#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
O=gpurun_out/r05/small_launch_blocks.txt
{ echo "# the shadow-ray kernel ALONE on a strip-sized launch (0.99 M rays = 1 per foreground pixel at 1600 x 1600; a strip of eight queues 0.86 M) against workgroups launched per CU; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for b in 1 2 3 4 6 8 12; do echo "blocks per CU $b: $(MIRRES_TRACE_BLOCKS_PER_CU=$b timeout 200 python3 scripts/dev_any_pmc.py 1600 1 20 0 2>&1 | tail -1)"; done
  echo "# the full-size launch (6.9 M rays) for scale"
  for b in 4 8; do echo "blocks per CU $b: $(MIRRES_TRACE_BLOCKS_PER_CU=$b timeout 200 python3 scripts/dev_any_pmc.py 1600 7 10 0 2>&1 | tail -1)"; done
  echo "# 566 x 566 x 7 rays (same count, a strip's coherence)"
  for b in 2 4 8; do echo "blocks per CU $b: $(MIRRES_TRACE_BLOCKS_PER_CU=$b timeout 200 python3 scripts/dev_any_pmc.py 566 7 20 0 2>&1 | tail -1)"; done
  echo "# lego-like, 1 ray per foreground pixel"
  for b in 2 4 8; do echo "blocks per CU $b: $(MIRRES_MESH=clustered MIRRES_TRACE_BLOCKS_PER_CU=$b timeout 200 python3 scripts/dev_any_pmc.py 1600 1 20 0 2>&1 | tail -1)"; done
} 2>&1 | tee $O
