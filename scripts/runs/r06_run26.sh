#!/bin/bash
# round 6, run 26: the N > 1 path of bench.py once more as a two-rank gloo dry run on one GPU, now with the `strip_exchange` record (the measured exchange
# time per sample in the bench line); both value schemes
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
for SCHEME in spp strips; do
MIRRES_DIST_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 2 --steps 2 --warmup 2 --spp 64 --no-roofline --value-scheme $SCHEME 2>/dev/null | tail -1 > gpurun_out/r06/r06_bench_two_ranks_gloo_dry_run_$SCHEME.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r06/r06_bench_two_ranks_gloo_dry_run_$SCHEME.json').read()); print('two ranks (gloo, one GPU), scheme $SCHEME: value', d['value'], d['config'].get('value_scheme'), 'strips', d.get('strips', {}).get('value'), 'spp', d.get('spp', {}).get('value')); print(d['config'].get('strip_balance')); print(d['config'].get('strip_exchange'))"
done
