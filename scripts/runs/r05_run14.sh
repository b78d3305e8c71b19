#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for p in normal low; do for cfg in "8 4 256 2" "2 1 128 2" "1 0 128 2"; do MIRRES_BULK_PRIO=$p python3 scripts/dev_strip_one.py $cfg 2>&1 | tail -1 | sed "s/^/bulk_prio=$p /"; done; done
for p in normal low; do MIRRES_BULK_PRIO=$p MIRRES_MESH=clustered python3 scripts/dev_strip_one.py 8 4 256 2 2>&1 | tail -1 | sed "s/^/clustered bulk_prio=$p /"; done
