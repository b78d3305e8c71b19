#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# ordered closest-hit kernel alone, icosphere: base = one pop per iteration, POP0 = pop until an entry survives (rounds 1-4), POP2 = two entries looked at per iteration"; bash scripts/dev_ab.sh 2 POP0 POP2
  echo "# lego-like"; MIRRES_MESH=clustered bash scripts/dev_ab.sh 2 POP0 POP2
  echo "# frames, icosphere"; bash scripts/dev_ab_frame.sh POP0 POP2
  echo "# frames, clustered"; MESH=clustered bash scripts/dev_ab_frame.sh POP0 POP2; } > gpurun_out/r05/ab_closest_pop.txt 2>&1
cat gpurun_out/r05/ab_closest_pop.txt
