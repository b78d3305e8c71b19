#!/bin/bash
# Round 4, GPU run 11: where the lego-like frame spends its time (kernel trace of bench.py --mesh clustered; three streams overlap, so the per-kernel times include sharing)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
rm -rf gpurun_out/ksc; timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ksc -o k -- python3 bench.py --mesh clustered --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r04/clustered_under_rocprof.log 2>&1
find gpurun_out/ksc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04/clustered_kernel_stats.csv
rm -rf gpurun_out/ksc
head -25 gpurun_out/r04/clustered_kernel_stats.csv | cut -c1-200
MIRRES_STREAMS=1 bash scripts/kstats.sh --mesh clustered --spp 128 --no-extras > gpurun_out/r04/clustered_kstats_serialised.txt 2>&1
MIRRES_STREAMS=1 bash scripts/kstats.sh --spp 128 --no-extras > gpurun_out/r04/icosphere_kstats_serialised.txt 2>&1
tail -26 gpurun_out/r04/clustered_kstats_serialised.txt | cut -c1-200; tail -26 gpurun_out/r04/icosphere_kstats_serialised.txt | cut -c1-200
