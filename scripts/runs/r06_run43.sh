#!/bin/bash
# round 6, run 43: the second invocation of run 42 printed no line (stderr was discarded): once more with stderr kept, another port, the watchdog at 45 s (twice: before and after bench.py kept the watchdog armed across the agreement)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
MIRRES_BENCH_WATCHDOG_S=45 MIRRES_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29763 bench.py --gpus 2 --steps 2 --warmup 2 --spp 64 --no-roofline --value-scheme strips > gpurun_out/r06/dry_strips.out 2> gpurun_out/r06/dry_strips.err
echo "rc=$?"; tail -1 gpurun_out/r06/dry_strips.out | cut -c1-600; grep -v "amdgpu.ids\|^$" gpurun_out/r06/dry_strips.err | tail -25 | cut -c1-300
