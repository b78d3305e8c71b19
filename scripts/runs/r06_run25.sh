#!/bin/bash
# round 6, run 25: full-size parity on the FINAL sources (after the straight-line pop and closest-hit pushes, csrc_sha e5ab1f759278): the metric's frame on the icosphere at 512 spp, the lego-like mesh at 128 spp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
SHA=$(python3 -c 'import bench; print(bench.csrc_sha())')
{ echo "csrc_sha $SHA"; timeout -k 10 1500 python3 scripts/dev_parity_big.py --res 1600 --spp 512 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_fullsize_512spp_parity_final2.txt; tail -2 gpurun_out/r06/r06_fullsize_512spp_parity_final2.txt
{ echo "csrc_sha $SHA"; MIRRES_MESH=clustered timeout -k 10 1200 python3 scripts/dev_parity_big.py --res 1600 --spp 128 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_clustered_fullsize_128spp_parity_final2.txt; tail -2 gpurun_out/r06/r06_clustered_fullsize_128spp_parity_final2.txt
