#!/bin/bash
# Round 4, GPU run 16: structural check of the private traversal layout (tests/test_gpu_layout.py), then the profile collection on the sources as they now are
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04 gpurun_out/out
rm -f gpurun_out/layout_check.txt
( time timeout 1500 python3 -m pytest tests/test_gpu_layout.py -m gpu -q ) 2>&1 | tail -30 | cut -c1-400
cat gpurun_out/layout_check.txt | cut -c1-300
bash scripts/profile_r04.sh 2>&1 | tail -40
# the counter snapshots just collected belong to these sources: bench.py reads them from profiles/
cp gpurun_out/out/pmc_any4q_summary.json gpurun_out/out/pmc_traffic.json profiles/
python3 bench.py --steps 5 --warmup 2 > gpurun_out/r04/bench_default_final.json 2> gpurun_out/r04/bench_default_final.err
python3 bench.py --mesh clustered --steps 5 --warmup 2 > gpurun_out/r04/bench_clustered.json 2> gpurun_out/r04/bench_clustered.err
tail -c 600 gpurun_out/r04/bench_clustered.err
