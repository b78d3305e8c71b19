#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
{ echo "# ordered closest-hit kernel alone, icosphere: base = pop until an entry survives, POP1 = one pop per iteration"; bash scripts/dev_ab.sh 2 POP1
  echo "# lego-like"; MIRRES_MESH=clustered bash scripts/dev_ab.sh 2 POP1; } 2>&1 | tee gpurun_out/r05/ab_closest_pop1.txt
