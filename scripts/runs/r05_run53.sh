#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ for mesh in icosphere clustered; do
    MIRRES_MESH=$mesh timeout 600 python3 scripts/dev_determinism_soak.py 32 40 2>&1 | grep "distinct"
    for st in 1 3 5; do MIRRES_MESH=$mesh MIRRES_STREAMS=$st timeout 300 python3 scripts/dev_determinism_soak.py 32 8 2>&1 | grep "distinct"; done
    MIRRES_MESH=$mesh MIRRES_PT_BATCH=5 timeout 300 python3 scripts/dev_determinism_soak.py 32 8 2>&1 | grep "distinct"
    MIRRES_MESH=$mesh MIRRES_TRACE_BLOCKS_PER_CU=3 timeout 300 python3 scripts/dev_determinism_soak.py 32 8 2>&1 | grep "distinct"
  done; } | tee gpurun_out/r05/determinism_soak.txt
