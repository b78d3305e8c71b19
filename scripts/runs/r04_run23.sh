#!/bin/bash
# Round 4, GPU run 23: key width of the position sort in front of the material lookup (MIRRES_GS_BITS 5 = default, 6, 7 bits per axis: 2 / 3 / 3 radix passes) and streams 2 vs 3, repeated
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_gs_bits.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do for mesh in icosphere clustered; do for b in 5 6 7; do echo "mesh $mesh gs_bits $b: $(MIRRES_GS_BITS=$b one $mesh)" >> $out; done; done; done
for mesh in icosphere clustered; do for b in 5 7; do echo "hash $mesh gs_bits $b: $(MIRRES_MESH=$mesh MIRRES_GS_BITS=$b python3 scripts/dev_frame_hash.py 12 2>/dev/null | tail -1)" >> $out; done; done
one512() { python3 bench.py --mesh $1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2 3; do for mesh in icosphere clustered; do for st in 2 3; do echo "mesh $mesh 512 spp streams $st: $(MIRRES_STREAMS=$st one512 $mesh)" >> $out; done; done; done
cat $out
