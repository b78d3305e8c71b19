#!/bin/bash
# round 6, run 14: (a) the N > 1 path of bench.py as a two-rank gloo dry run on one GPU (declared value scheme, the balancer's busy times); (b) k_spatial_resolve forced to
# four / five waves per SIMD where the chain IS the period: the training step and a strip of eight (VERDICT r5 item 5 asked for the small-frame cases)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
MIRRES_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 2 --steps 2 --warmup 2 --spp 32 --no-roofline 2>/dev/null | tail -1 > gpurun_out/r06/r06_bench_two_ranks_gloo_dry_run.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r06/r06_bench_two_ranks_gloo_dry_run.json').read()); print('two ranks (gloo, one GPU): value', d['value'], d['config'].get('value_scheme'), 'strips', d.get('strips', {}).get('value'), d['config'].get('strip_balance'))"
O=gpurun_out/r06/ab_resolve_waves.txt
{ echo "# k_spatial_resolve<5, true> with amdgpu_waves_per_eu forced (default: the allocator's choice, 164 registers = 3 waves per SIMD); csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for v in base sres4 sres5 base sres4 sres5; do
    if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
    echo "$v  training step: $(timeout 300 python3 scripts/train_step_bench.py --steps 6 2>&1 | grep '^stage-1' | cut -c29-45)   strip 4 of 8, 256 spp: $(timeout 300 python3 scripts/dev_strip_one.py 8 4 256 3 2>&1 | tail -1 | cut -c1-120)"
  done
} 2>&1 | tee $O
SHA=$(python3 -c 'import bench; print(bench.csrc_sha())')
{ echo "csrc_sha $SHA"; timeout -k 10 1500 python3 scripts/dev_parity_big.py --res 1600 --spp 256 --env 1024x2048 --albedo_scale 0.9,0.8,0.7 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06/r06_configs3_256spp_parity.txt; tail -2 gpurun_out/r06/r06_configs3_256spp_parity.txt
