#!/bin/bash
# round 6, run 15: environment-table row scans through LDS tiles (same sums, same order): tests, kernel time, training step
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_passes.py tests/test_gpu_fullsize.py -x -q -k "passes or configs3 or one_sample" 2>&1 | tail -4
rm -rf gpurun_out/pf; mkdir -p gpurun_out/pf
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o t -- python3 scripts/train_step_bench.py --steps 3 > gpurun_out/pf/log_tr 2>&1
f=$(find gpurun_out/pf/kt -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'k_env' in r['Name']: print(r['Name'][:40], r['Calls'], '%.1f us' % (float(r['AverageNs'])/1e3))
"
rm -rf gpurun_out/pf
for i in 1 2 3; do timeout 300 python3 scripts/train_step_bench.py --steps 6 2>&1 | grep '^stage-1' | cut -c1-60; done
