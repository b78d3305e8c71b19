#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# leaf branch of k_trace_any4q: counters of the frame's own rays (scripts/dev_leaf_branch.py 4)"; python3 scripts/dev_leaf_branch.py 4 2>&1 | grep -v amdgpu.ids
  for v in P8 P16 P24; do echo "## with -DMR_ANY_PARK=${v#P}"; MIRRES_LIB=$PWD/ab/libmirres_$v.so python3 scripts/dev_leaf_branch.py 4 2>&1 | grep -v amdgpu.ids; done
  echo "# shadow-ray kernel alone, icosphere (scripts/dev_any_pmc.py 1600 7 10): base = every lane tests its leaf at once"; bash scripts/dev_ab.sh 0 P8 P16 P24
  echo "# the same, lego-like mesh"; MIRRES_MESH=clustered bash scripts/dev_ab.sh 0 P8 P16 P24
} > gpurun_out/r05/ab_park.txt 2>&1
cat gpurun_out/r05/ab_park.txt
