#!/bin/bash
# Round 5, GPU run 1: counters of every kernel of the per-sample chain (both meshes) + the strip-scheme scaling table (both meshes)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
bash scripts/pmc_chain.sh chain_ico icosphere 8 > gpurun_out/r05/pmc_chain_icosphere.txt 2>&1
cp gpurun_out/pmc_chain_ico/summary.json gpurun_out/r05/pmc_chain_icosphere.json
bash scripts/pmc_chain.sh chain_clu clustered 8 > gpurun_out/r05/pmc_chain_clustered.txt 2>&1
cp gpurun_out/pmc_chain_clu/summary.json gpurun_out/r05/pmc_chain_clustered.json
timeout -k 5 600 python3 scripts/dev_strip_table.py 64 2 > gpurun_out/r05/strip_table_icosphere.txt 2>&1
MIRRES_MESH=clustered timeout -k 5 600 python3 scripts/dev_strip_table.py 64 2 > gpurun_out/r05/strip_table_clustered.txt 2>&1
head -30 gpurun_out/r05/pmc_chain_icosphere.txt | cut -c1-200
tail -25 gpurun_out/r05/strip_table_icosphere.txt
tail -8 gpurun_out/r05/strip_table_clustered.txt
