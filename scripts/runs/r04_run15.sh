#!/bin/bash
# Round 4, GPU run 15: the driver's own sequence on the final tree (smoke, default bench with the driver's step counts) + the standalone lego-like bench line
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04/bench_driver_style.json 2> gpurun_out/r04/bench_driver_style.err
tail -4 gpurun_out/r04/bench_driver_style.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r04/bench_driver_style.json').read().strip().splitlines()[-1]); r=d['roofline']
print('driver style:', d['value'], d['ms_per_step'], 'bound', r['bound'], 'frac', r['frac'], 'clustered', d['clustered']['value'], 'train', d['train_step']['ms_per_step'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['max_abs_err'])"
python3 bench.py --mesh clustered --steps 5 --warmup 2 > gpurun_out/r04/bench_clustered.json 2> gpurun_out/r04/bench_clustered.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r04/bench_clustered.json').read().strip().splitlines()[-1]); r=d['roofline']
print('clustered standalone:', d['value'], d['ms_per_step'], d['config']['foreground_frac'], r['launch_ms'], r['grays_per_s'], r['per_ray'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['max_abs_err'], 'icosphere sub-record', d['icosphere']['value'])"
( time timeout 600 python3 -m pytest tests/test_gpu_clustered.py -m gpu -q -k "installed" ) 2>&1 | tail -4
