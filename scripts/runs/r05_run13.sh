#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for cfg in "8 4 256 2 bg" "8 4 256 2"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf gpurun_out/kt; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o k -- python3 scripts/dev_strip_one.py $cfg > gpurun_out/r05/strip2_prof_$tag.txt 2>&1
  find gpurun_out/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05/strip2_kstats_$tag.csv
done
rm -rf gpurun_out/kt
for b in 2 3 4 6 8; do MIRRES_TRACE_BLOCKS_PER_CU=$b python3 scripts/dev_strip_one.py 8 4 256 2 2>&1 | tail -1 | sed "s/^/blocks_per_cu=$b /"; done
for b in 4 6 8; do MIRRES_TRACE_BLOCKS_PER_CU=$b python3 scripts/dev_strip_one.py 2 1 128 2 2>&1 | tail -1 | sed "s/^/blocks_per_cu=$b /"; done
