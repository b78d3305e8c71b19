#!/bin/bash
# Round 4, GPU run 9: builder-run long parity frames on the final kernels (the oracle on the box's 128 cores is the checker): the metric's own frame, and the lego-like mesh at full size
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time python3 scripts/dev_parity_big.py --res 1600 --spp 512 ) > gpurun_out/r04/fullsize_512spp_parity.txt 2>&1
tail -9 gpurun_out/r04/fullsize_512spp_parity.txt
( time MIRRES_MESH=clustered python3 scripts/dev_parity_big.py --res 1600 --spp 96 ) > gpurun_out/r04/clustered_fullsize_96spp_parity.txt 2>&1
tail -9 gpurun_out/r04/clustered_fullsize_96spp_parity.txt
python3 bench.py > gpurun_out/r04/bench_default_with_snapshot.json 2> gpurun_out/r04/bench_default_with_snapshot.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r04/bench_default_with_snapshot.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], 'hbm_counter', r['hbm_counter'], 'snapshot', {k:(v.get('source') if isinstance(v,dict) else v) for k,v in r['pmc_snapshot'].items()})"
