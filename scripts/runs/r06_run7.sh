#!/bin/bash
# round 6, run 7: backward kernels with sector-grouped atomics — gradient tests, training-step time, kernel trace
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_adjoints.py tests/test_gpu_matnet.py tests/test_gpu_render.py -x -q 2>&1 | tail -6 | tee gpurun_out/r06/tests_run7.txt
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "training" 2>&1 | tail -3 | tee -a gpurun_out/r06/tests_run7.txt
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/pf; mkdir -p gpurun_out/pf
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o t -- python3 scripts/train_step_bench.py --steps 3 > gpurun_out/pf/log_tr 2>&1
find gpurun_out/pf/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06/train_step_kernel_stats_b.csv; grep '^stage-1' gpurun_out/pf/log_tr | tee gpurun_out/r06/train_step_b.txt
rm -rf gpurun_out/pf
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r06/train_step_kernel_stats_b.csv')):
    if 'bwd' in r['Name'] or float(r['Percentage']) > 3: print("%-60s %4s calls %10.1f us avg %6s %%" % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
for i in 1 2 3; do timeout 300 python3 scripts/train_step_bench.py --steps 6 2>&1 | grep '^stage-1' | cut -c1-90; done | tee gpurun_out/r06/train_step_b_times.txt
