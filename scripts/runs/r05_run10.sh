#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
bash scripts/pmc_any.sh g3 1600 7 3 0 > gpurun_out/r05/pmc_any_mode3.txt 2>&1
MIRRES_LIB=$PWD/ab/libmirres_GRAB0.so bash scripts/pmc_any.sh g0 1600 7 3 0 > gpurun_out/r05/pmc_any_mode0.txt 2>&1
paste <(grep -A60 "^k_trace_any4q" gpurun_out/r05/pmc_any_mode0.txt | head -50) <(grep -A60 "^k_trace_any4q" gpurun_out/r05/pmc_any_mode3.txt | head -50 | awk '{print $2}')
