#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout -k 5 1200 python3 -m pytest tests/test_gpu_render.py -m gpu -x -q -k "strip" > gpurun_out/r05/gputests_strip.txt 2>&1; tail -3 gpurun_out/r05/gputests_strip.txt
for m in icosphere clustered; do
  MIRRES_MESH=$m timeout -k 5 900 python3 scripts/dev_strip_table.py 128 2 default 2,4,8 4 > gpurun_out/r05/strip_table4_$m.txt 2>&1
  grep -E "^N=|^fit|whole frame" gpurun_out/r05/strip_table4_$m.txt
done
