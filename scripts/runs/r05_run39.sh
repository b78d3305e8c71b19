#!/bin/bash
# Round 5, GPU run 39: the metric's own frame (1600 x 1600 x 512 spp) on the FINAL kernels (csrc_sha 296dc92daef1) against the CPU oracle, bit for bit
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
python3 -c "import bench; print('csrc_sha', bench.csrc_sha())" > gpurun_out/r05/fullsize_512spp_parity.txt
timeout -k 5 2400 python3 scripts/dev_parity_big.py --res 1600 --spp 512 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05/fullsize_512spp_parity.txt; tail -3 gpurun_out/r05/fullsize_512spp_parity.txt
