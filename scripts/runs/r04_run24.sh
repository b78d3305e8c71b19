#!/bin/bash
# Round 4, GPU run 24: 8 bits per axis for the position sort (same three passes as 7); streams 2 vs 3 on the training step
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_gs_bits8.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do for mesh in icosphere clustered; do for b in 7 8; do echo "mesh $mesh gs_bits $b: $(MIRRES_GS_BITS=$b one $mesh)" >> $out; done; done; done
for rep in 1 2; do for st in 2 3; do echo "train step streams $st: $(MIRRES_STREAMS=$st python3 scripts/train_step_bench.py 2>/dev/null | tail -1)" >> $out; done; done
cat $out
