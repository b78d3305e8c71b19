#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
one() { timeout -k 5 300 python3 bench.py --mesh $1 --no-extras --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
{ echo "# light-tile records of the batched initial resampling: 16-byte (direction re-derived per candidate; default since round 4) against 32-byte (MIRRES_TILE_COMPACT=0), re-measured on the VALU-bound frame of round 5; 128 spp"
  for rep in 1 2; do for m in icosphere clustered; do echo "$m compact   $(one $m)"; echo "$m 32-byte   $(MIRRES_TILE_COMPACT=0 one $m)"; done; done; } | tee gpurun_out/r05/ab_tile_compact.txt
