#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# kernel alone: base = mode 3 (sc1 load of the mask), GRAB0 = blind, M4 = mask by plain load, M5 = mask by nontemporal load, S4 / S8 = blind, stealing from the next 4 / 8 sub-queues only"
  bash scripts/dev_ab.sh 0 GRAB0 M4 M5 S4 S8
  echo "# background-only strip / strip 4 of 8 (256 spp)"
  for v in GRAB0 M4 M5 S4 S8; do for cfg in "8 4 256 2 bg" "8 4 256 2"; do MIRRES_LIB=$PWD/ab/libmirres_$v.so python3 scripts/dev_strip_one.py $cfg 2>&1 | tail -1 | sed "s/^/$v /"; done; done
} > gpurun_out/r05/ab_grab_more.txt 2>&1
cat gpurun_out/r05/ab_grab_more.txt
