#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout -k 5 900 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_rccl.py -m gpu -x -q -k "strip or rccl or shard or rank" 2>&1 | tail -3
MIRRES_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --spp 16 --steps 2 --warmup 2 --no-roofline > gpurun_out/r05/bench_two_ranks_gloo.json 2> gpurun_out/r05/bench_two_ranks_gloo.err; tail -c 700 gpurun_out/r05/bench_two_ranks_gloo.json
