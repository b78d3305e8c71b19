#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
bash scripts/pmc_chain.sh chain_ico icosphere 8 > gpurun_out/r05/pmc_chain2_icosphere.txt 2>&1
head -14 gpurun_out/r05/pmc_chain2_icosphere.txt | cut -c1-260
