#!/bin/bash
# Round 4, GPU run 19: the whole -m gpu suite on the final sources, the profile collection on them, then the bench lines (default, clustered, driver style) with the fresh snapshots
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04 gpurun_out/out
export MIRRES_PARITY_REPORT=$PWD/gpurun_out/r04/parity_report.txt; rm -f $MIRRES_PARITY_REPORT gpurun_out/clustered_mesh_report.txt gpurun_out/layout_check.txt
( time timeout 2400 python3 -m pytest tests -m gpu -q ) > gpurun_out/r04/gpu_tests.log 2>&1
unset MIRRES_PARITY_REPORT
tail -6 gpurun_out/r04/gpu_tests.log | cut -c1-300
bash scripts/profile_r04.sh 2>&1 | tail -12
cp gpurun_out/out/pmc_any4q_summary.json gpurun_out/out/pmc_traffic.json profiles/
python3 bench.py --steps 5 --warmup 2 > gpurun_out/r04/bench_default_final.json 2> gpurun_out/r04/bench_default_final.err
python3 bench.py --mesh clustered --steps 5 --warmup 2 > gpurun_out/r04/bench_clustered.json 2> gpurun_out/r04/bench_clustered.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_style.json 2> gpurun_out/r04/bench_driver_style.err
for f in bench_default_final bench_clustered bench_driver_style; do python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r04/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], r['bound'], r['frac'], r['launch_ms'], r['grays_per_s'], r['rays_not_traced_frac'], d.get('parity_failed'))"; done
