#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
export MIRRES_TRACE_BLOCKS_PER_CU=6
{ echo "## phases, MR_GRAB_MODE 0 (6 workgroups per CU: the instrumentation buffer holds 6144 waves)"; MIRRES_LIB=$PWD/ab/libmirres_PH0.so python3 scripts/dev_phases.py 7 2>&1 | grep -v amdgpu.ids
  echo "## phases, MR_GRAB_MODE 3"; MIRRES_LIB=$PWD/ab/libmirres_PH3.so python3 scripts/dev_phases.py 7 2>&1 | grep -v amdgpu.ids
  echo "## kernel alone at 6 workgroups per CU"; bash scripts/dev_ab.sh 0 GRAB0; } > gpurun_out/r05/phases_grab.txt 2>&1
cat gpurun_out/r05/phases_grab.txt
