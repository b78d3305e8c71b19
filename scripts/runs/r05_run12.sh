#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
for m in icosphere clustered; do for w in default 0.0; do
  MIRRES_MESH=$m timeout -k 5 600 python3 scripts/dev_strip_table.py 128 2 $w > gpurun_out/r05/strip_table3_${m}_$w.txt 2>&1
  grep -E "^N=|^fit|whole frame" gpurun_out/r05/strip_table3_${m}_$w.txt
done; done
