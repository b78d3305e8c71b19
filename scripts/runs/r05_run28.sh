#!/bin/bash
# Round 5, GPU run 28: builder-run long frames on the final kernels (csrc_sha e62d2db77dc9): HIP path vs the CPU oracle, bit for bit
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout -k 5 1500 python3 scripts/dev_parity_big.py --res 1600 --spp 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/fullsize_256spp_parity.txt; tail -4 gpurun_out/r05/fullsize_256spp_parity.txt
MIRRES_MESH=clustered timeout -k 5 900 python3 scripts/dev_parity_big.py --res 1600 --spp 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/clustered_fullsize_64spp_parity.txt; tail -4 gpurun_out/r05/clustered_fullsize_64spp_parity.txt
timeout -k 5 900 python3 scripts/dev_parity_big.py --res 1024 --spp 128 --bounces 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/configs4_128spp_parity.txt; tail -4 gpurun_out/r05/configs4_128spp_parity.txt
