#!/bin/bash
# Round 4, GPU run 22: knobs re-swept on the final kernels — samples per batch (MIRRES_PT_BATCH) and stream count (MIRRES_STREAMS), both meshes, the metric's 512 spp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/knobs_final.txt; : > $out
one() { python3 bench.py --mesh $1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for mesh in icosphere clustered; do
  for k in 16 24 32 48 64; do echo "mesh $mesh batch $k streams 3: $(MIRRES_PT_BATCH=$k one $mesh)" >> $out; done
  for st in 2 4 5; do echo "mesh $mesh batch 32 streams $st: $(MIRRES_STREAMS=$st one $mesh)" >> $out; done
  echo "mesh $mesh batch 32 streams 3 (again): $(one $mesh)" >> $out
done
cat $out
