#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout -k 10 600 python3 bench.py > gpurun_out/r05/bench_final.json 2> gpurun_out/r05/bench_final.err
echo "rc=$? lines=$(wc -l < gpurun_out/r05/bench_final.json)"; python3 -c "
import json; d=json.loads(open('gpurun_out/r05/bench_final.json').read().strip())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['clustered']['value'], d['train_step']['ms_per_step'])"
