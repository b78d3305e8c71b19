#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_render.py -m gpu -x -q -k "strip" 2>&1 | tail -4
{ echo "# csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for w in 8 4 2; do timeout 600 python3 scripts/dev_strip_overlap.py $w 44 256 2>&1 | grep -v amdgpu.ids; done
  MIRRES_MESH=clustered timeout 600 python3 scripts/dev_strip_overlap.py 8 44 256 2>&1 | grep -v amdgpu.ids; } | tee gpurun_out/r05/strip_overlap_resolve.txt
