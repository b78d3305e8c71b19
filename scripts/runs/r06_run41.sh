#!/bin/bash
# round 6, run 41: the metric's frame (icosphere, 512 spp) against the CPU oracle on the tree the round ends with (csrc_sha 70ecfb50543c: 64 samples per batch)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
SHA=$(python3 -c 'import bench; print(bench.csrc_sha())')
{ echo "csrc_sha $SHA"; timeout -k 10 1500 python3 scripts/dev_parity_big.py --res 1600 --spp 512 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_fullsize_512spp_parity_final3.txt; tail -2 gpurun_out/r06/r06_fullsize_512spp_parity_final3.txt
