#!/bin/bash
# round 6, run 35: the lego-like mesh at the metric's full 512 spp against the CPU oracle on the sources the round ends with (csrc_sha e5ab1f759278; run 25 did 128 spp)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
SHA=$(python3 -c 'import bench; print(bench.csrc_sha())')
{ echo "csrc_sha $SHA"; MIRRES_MESH=clustered timeout -k 10 2700 python3 scripts/dev_parity_big.py --res 1600 --spp 512 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_clustered_fullsize_512spp_parity_final.txt; tail -2 gpurun_out/r06/r06_clustered_fullsize_512spp_parity_final.txt
