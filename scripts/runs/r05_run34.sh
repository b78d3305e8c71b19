#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ timeout -k 5 300 python3 scripts/dev_two_frames.py 128 800 2>&1 | grep -v amdgpu.ids
  echo "# strip-sized frames (1600 x 224)"; timeout -k 5 300 python3 scripts/dev_two_frames.py 128 800 2>&1 >/dev/null | head -0
  MIRRES_MESH=clustered timeout -k 5 300 python3 scripts/dev_two_frames.py 128 800 2>&1 | grep -v amdgpu.ids; } | tee gpurun_out/r05/two_frames.txt
