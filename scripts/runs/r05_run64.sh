#!/bin/bash
# dry run of bench.py's N > 1 path on ONE GPU: two ranks over gloo sharing cuda:0 (RCCL refuses two ranks on one device), small frame; checks the control flow of both schemes, the balancer, the scheme choice and the line
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
MIRRES_DIST_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 2 --spp 16 --res 400 > gpurun_out/r05/bench_two_ranks_gloo.json 2> gpurun_out/r05/bench_two_ranks_gloo.err
echo "rc=$?"; wc -l gpurun_out/r05/bench_two_ranks_gloo.json; cut -c1-1500 gpurun_out/r05/bench_two_ranks_gloo.json; tail -5 gpurun_out/r05/bench_two_ranks_gloo.err
