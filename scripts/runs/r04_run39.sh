#!/bin/bash
# Round 4, GPU run 39: kernel trace of the lego-like frame on the final state (the icosphere's is profiles/r04_kernel_stats.csv)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/kc; timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kc -o k -- python3 bench.py --mesh clustered --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r04/clustered_under_rocprof.json 2> gpurun_out/r04/clustered_under_rocprof.err
find gpurun_out/kc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04/clustered_kernel_stats_final.csv
rm -rf gpurun_out/kc
head -14 gpurun_out/r04/clustered_kernel_stats_final.csv | cut -c1-120
tail -c 300 gpurun_out/r04/clustered_under_rocprof.json
