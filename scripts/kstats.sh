#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
rm -rf gpurun_out/kt; mkdir -p gpurun_out/kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o kt -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline "$@" > gpurun_out/kt/bench.log 2>&1
grep '^{' gpurun_out/kt/bench.log | tail -1 | cut -c1-220
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/kt/kt_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:22]:
    print("%6.2f%% %8.1f us x%5s  %s" % (float(r['Percentage']), float(r['AverageNs'])/1e3, r['Calls'], r['Name'][:90]))
PY
rm -f gpurun_out/kt/kt_kernel_trace.csv
