#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python3 scripts/dev_strip_table.py 512 1 default 2,4,8 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/strip_table_512_icosphere.txt
MIRRES_MESH=clustered timeout 900 python3 scripts/dev_strip_table.py 512 1 default 2,4,8 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/strip_table_512_clustered.txt
grep "^N=\|whole frame\|^fit" gpurun_out/r05/strip_table_512_*.txt
