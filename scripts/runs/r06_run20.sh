#!/bin/bash
# round 6, run 20: run-to-run determinism soak on the final sources — six schedules / hierarchies / band settings per mesh, every output buffer of every frame hashed
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
{ for mesh in icosphere clustered; do
    MIRRES_MESH=$mesh timeout 600 python3 scripts/dev_determinism_soak.py 32 40 2>&1 | grep "distinct"
    for st in 1 3 5; do MIRRES_MESH=$mesh MIRRES_STREAMS=$st timeout 300 python3 scripts/dev_determinism_soak.py 32 8 2>&1 | grep "distinct"; done
    MIRRES_MESH=$mesh MIRRES_PT_BATCH=5 timeout 300 python3 scripts/dev_determinism_soak.py 32 8 2>&1 | grep "distinct"
    MIRRES_MESH=$mesh MIRRES_PRIVATE_TREE=2 timeout 300 python3 scripts/dev_determinism_soak.py 32 8 2>&1 | grep "distinct" | sed 's/$/  [MIRRES_PRIVATE_TREE=2]/'
    MIRRES_MESH=$mesh MIRRES_BANDS=4 MIRRES_CHAIN_STREAMS=3 timeout 300 python3 scripts/dev_determinism_soak.py 32 8 2>&1 | grep "distinct" | sed 's/$/  [MIRRES_BANDS=4 MIRRES_CHAIN_STREAMS=3]/'
  done; } | tee gpurun_out/r06/r06_determinism_soak.txt
