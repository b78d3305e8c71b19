#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/gather_coop.hip -o /tmp/gather_coop && timeout -k 3 120 /tmp/gather_coop 2>&1 | tee gpurun_out/r05/gather_coop.txt
