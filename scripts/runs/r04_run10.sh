#!/bin/bash
# Round 4, GPU run 10: the N > 1 path of bench.py after this round's refactoring — two ranks over gloo on the one GPU (dry run; RCCL refuses two ranks on one device)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
MIRRES_DIST_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --spp 16 --steps 1 --warmup 1 > gpurun_out/r04/bench_two_ranks_gloo_dry_run.json 2> gpurun_out/r04/bench_two_ranks_gloo_dry_run.err
tail -3 gpurun_out/r04/bench_two_ranks_gloo_dry_run.err | cut -c1-300
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/bench_two_ranks_gloo_dry_run.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], d['config']['parallelism'], d.get('strips'))"
( time timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "round4_switches" ) > gpurun_out/r04/gpu_tests_switches.log 2>&1
tail -5 gpurun_out/r04/gpu_tests_switches.log | cut -c1-300
