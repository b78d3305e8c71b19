#!/bin/bash
# round 6, run 6: fp32 atomic-add rates by scope / contention / lane grouping (scripts/ubench/atomic_rate.hip)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/atomic_rate.hip -o /tmp/atomic_rate 2>/dev/null && timeout 300 /tmp/atomic_rate 2>&1 | tee gpurun_out/r06/atomic_rate.txt
