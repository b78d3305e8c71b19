#!/bin/bash
# round 6, run 27: configs[4] (1024^2, 512 spp, 3 bounces) and configs[3] (relighting: 1024x2048 environment, albedo scale; 256 spp) against the CPU oracle on the FINAL
# sources (csrc_sha e5ab1f759278), then the two-rank dry run of bench.py (run 26's commands)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
SHA=$(python3 -c 'import bench; print(bench.csrc_sha())')
{ echo "csrc_sha $SHA"; timeout -k 10 1200 python3 scripts/dev_parity_big.py --res 1024 --spp 512 --bounces 3 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_configs4_512spp_parity_final.txt; tail -2 gpurun_out/r06/r06_configs4_512spp_parity_final.txt
{ echo "csrc_sha $SHA"; timeout -k 10 1200 python3 scripts/dev_parity_big.py --res 1600 --spp 256 --env 1024x2048 --albedo_scale 0.9,0.8,0.7 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_configs3_256spp_parity_final.txt; tail -2 gpurun_out/r06/r06_configs3_256spp_parity_final.txt
bash scripts/runs/r06_run26.sh
