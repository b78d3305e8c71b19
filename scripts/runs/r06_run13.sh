#!/bin/bash
# round 6, run 13: full-size parity of the final sources against the CPU oracle — the metric's own frame on both meshes, configs[3] and configs[4] at their one-GPU size
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
SHA=$(python3 -c 'import bench; print(bench.csrc_sha())')
{ echo "csrc_sha $SHA"; timeout -k 10 1500 python3 scripts/dev_parity_big.py --res 1600 --spp 512 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_fullsize_512spp_parity.txt; tail -2 gpurun_out/r06/r06_fullsize_512spp_parity.txt
{ echo "csrc_sha $SHA"; MIRRES_MESH=clustered timeout -k 10 2400 python3 scripts/dev_parity_big.py --res 1600 --spp 512 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_clustered_fullsize_512spp_parity.txt; tail -2 gpurun_out/r06/r06_clustered_fullsize_512spp_parity.txt
{ echo "csrc_sha $SHA"; timeout -k 10 1200 python3 scripts/dev_parity_big.py --res 1024 --spp 512 --bounces 3 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_configs4_512spp_parity.txt; tail -2 gpurun_out/r06/r06_configs4_512spp_parity.txt
{ echo "csrc_sha $SHA"; timeout -k 10 1800 python3 scripts/dev_parity_big.py --res 1600 --spp 512 --env 1024x2048 --albedo_scale 0.9,0.8,0.7 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_configs3_512spp_parity.txt; tail -2 gpurun_out/r06/r06_configs3_512spp_parity.txt
