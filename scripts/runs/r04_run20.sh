#!/bin/bash
# Round 4, GPU run 20: the long parity frames on the FINAL kernels (after the unseen-ray rule and the record swizzle): the metric's own frame on the icosphere, the lego-like mesh at
# full size with 96 samples; then the N > 1 path of bench.py as a two-rank gloo dry run on the one GPU
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time MIRRES_MESH=clustered python3 scripts/dev_parity_big.py --res 1600 --spp 96 ) > gpurun_out/r04/clustered_fullsize_96spp_parity.txt 2>&1
tail -12 gpurun_out/r04/clustered_fullsize_96spp_parity.txt | cut -c1-300
( time python3 scripts/dev_parity_big.py --res 1600 --spp 512 ) > gpurun_out/r04/fullsize_512spp_parity.txt 2>&1
tail -12 gpurun_out/r04/fullsize_512spp_parity.txt | cut -c1-300
MIRRES_DIST_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --spp 16 --steps 1 --warmup 1 > gpurun_out/r04/bench_two_ranks_gloo_dry_run.json 2> gpurun_out/r04/bench_two_ranks_gloo_dry_run.err
tail -3 gpurun_out/r04/bench_two_ranks_gloo_dry_run.err | cut -c1-300
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/bench_two_ranks_gloo_dry_run.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], d['config']['parallelism'], d.get('strips'))"
