#!/bin/bash
# Round 5, GPU run 2: where does a strip's per-sample fixed cost go?  kernel traces of strip 4 of 8, a background-only strip and the whole frame
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
for cfg in "8 4 64 3" "8 4 64 3 bg" "1 0 64 3" "8 4 256 2" "8 4 256 2 bg"; do
  tag=$(echo $cfg | tr ' ' '_')
  python3 scripts/dev_strip_one.py $cfg > gpurun_out/r05/strip_one_$tag.txt 2>&1
  rm -rf gpurun_out/kt; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o k -- python3 scripts/dev_strip_one.py $cfg > gpurun_out/r05/strip_one_prof_$tag.txt 2>&1
  find gpurun_out/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05/strip_one_kstats_$tag.csv
  tail -1 gpurun_out/r05/strip_one_$tag.txt
done
rm -rf gpurun_out/kt
