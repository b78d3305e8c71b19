#!/bin/bash
# the metric's frame (1600 x 1600 internal, 512 spp) on the FINAL kernel sources against the CPU oracle, both meshes (the lego-like one at its full 512 spp for the first time)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
SHA=$(python3 -c 'import bench; print(bench.csrc_sha())')
{ echo "csrc_sha $SHA"; timeout -k 10 1500 python3 scripts/dev_parity_big.py --res 1600 --spp 512 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05/fullsize_512spp_parity_final.txt
{ echo "csrc_sha $SHA"; MIRRES_MESH=clustered timeout -k 10 2400 python3 scripts/dev_parity_big.py --res 1600 --spp 512 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05/clustered_fullsize_512spp_parity_final.txt
tail -n 3 gpurun_out/r05/fullsize_512spp_parity_final.txt; tail -n 3 gpurun_out/r05/clustered_fullsize_512spp_parity_final.txt
