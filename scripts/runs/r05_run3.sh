#!/bin/bash
# Round 5, GPU run 3: probe-before-atomic queue heads (grab_chunk) — GPU tests, A/B against the blind-atomic build on both meshes, strip table again
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout -k 5 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05/gputests_run3.txt 2>&1; tail -3 gpurun_out/r05/gputests_run3.txt
{ echo "# scripts/dev_ab_frame.sh BLIND (bench.py --spp 128 --steps 3 --warmup 1, interleaved, two rounds); base = probe-before-atomic queue heads, BLIND = -DMR_EXP_BLIND_GRAB (rounds 1-4)"
  echo "## icosphere"; bash scripts/dev_ab_frame.sh BLIND
  echo "## clustered"; MESH=clustered bash scripts/dev_ab_frame.sh BLIND; } > gpurun_out/r05/ab_grab.txt 2>&1
cat gpurun_out/r05/ab_grab.txt
timeout -k 5 600 python3 scripts/dev_strip_table.py 128 2 0.0 > gpurun_out/r05/strip_table2_icosphere.txt 2>&1
tail -6 gpurun_out/r05/strip_table2_icosphere.txt
for cfg in "8 4 256 2 bg" "8 4 256 2" "1 0 256 2"; do python3 scripts/dev_strip_one.py $cfg 2>&1 | tail -1; done
