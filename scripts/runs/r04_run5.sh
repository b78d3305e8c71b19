#!/bin/bash
# Round 4, GPU run 5: hostile shading inputs vs the oracle, hash-grid locality experiment on both meshes, deep-tree test with the private hierarchy
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
export MIRRES_PARITY_REPORT=$PWD/gpurun_out/r04/parity_hostile.txt; rm -f $MIRRES_PARITY_REPORT
( time timeout 900 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_bvh.py -m gpu -q -k "hostile or deep_tree or degenerate" ) > gpurun_out/r04/gpu_tests_hostile.log 2>&1
unset MIRRES_PARITY_REPORT
tail -30 gpurun_out/r04/gpu_tests_hostile.log | cut -c1-400
cat gpurun_out/r04/parity_hostile.txt | cut -c1-200
for mesh in icosphere clustered; do echo "== $mesh"; MIRRES_MESH=$mesh python3 scripts/dev_grid_locality.py 4 2>&1 | tail -8; done > gpurun_out/r04/grid_locality.txt
cat gpurun_out/r04/grid_locality.txt
