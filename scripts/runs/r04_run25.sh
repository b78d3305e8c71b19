#!/bin/bash
# Round 4, GPU run 25: what ray order is worth on the rays of the second indirect vertex (scripts/dev_sort_bounce_rays.py), both meshes
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/sort_bounce_rays.txt; : > $out
for mesh in icosphere clustered; do MIRRES_MESH=$mesh timeout 600 python3 scripts/dev_sort_bounce_rays.py 4 2>&1 | grep -v amdgpu.ids >> $out; done
cat $out
