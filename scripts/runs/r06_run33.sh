#!/bin/bash
# round 6, run 33: the round's last check of the tree as it is committed: smoke(), the whole GPU suite, the driver's bench command (no counter passes: the snapshots of the
# third collection stay fresh — csrc unchanged)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
python3 -c "import __graft_entry__ as G; G.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06/gpu_suite_d.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06/r06_bench_driver_style_last.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r06/r06_bench_driver_style_last.json').read()); print(d['value'], d['ms_per_step'], d.get('clustered',{}).get('value'), d.get('train_step',{}).get('ms_per_step'), 'roofline', d['roofline']['frac'], 'stale' , d['roofline'].get('pmc_stale'))"
