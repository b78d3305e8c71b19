#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05 gpurun_out/pf
for mode in "behind the resolve" "exposed"; do
  rm -rf gpurun_out/pf/kt
  ONLY="$mode" timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pf/kt -o kt -- python3 scripts/dev_strip_overlap.py 8 44 64 > gpurun_out/pf/log 2>&1
  echo "## $mode"; grep "per sample" gpurun_out/pf/log
  python3 scripts/dev_trace_window.py "$(find gpurun_out/pf/kt -name '*kernel_trace.csv' | head -1)" 0.7 40 "spatial|any4q<false, 0, 0, 1>|sleep|spin|temporal"
done 2>&1 | tee gpurun_out/r05/strip_overlap_window.txt
rm -rf gpurun_out/pf
