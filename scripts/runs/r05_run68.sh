#!/bin/bash
# BASELINE configs[4]'s one-GPU shape (1024 x 1024, 512 spp, 3 indirect bounces) and configs[3]'s (800 x 800 at ssaa 2, 512 spp, external 1024 x 2048 map, albedo scale) on the final sources against the CPU oracle
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
SHA=$(python3 -c 'import bench; print(bench.csrc_sha())')
{ echo "csrc_sha $SHA"; timeout -k 10 1200 python3 scripts/dev_parity_big.py --res 1024 --spp 512 --bounces 3 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05/configs4_512spp_parity_final.txt
{ echo "csrc_sha $SHA"; timeout -k 10 1800 python3 scripts/dev_parity_big.py --res 1600 --spp 512 --env 1024x2048 --albedo_scale 0.9,0.8,0.7 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05/configs3_512spp_parity_final.txt
tail -n 2 gpurun_out/r05/configs4_512spp_parity_final.txt; tail -n 2 gpurun_out/r05/configs3_512spp_parity_final.txt
