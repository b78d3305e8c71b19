#!/bin/bash
# round 6, run 38: the runtime knobs of mirres_render re-checked on the final kernels at the metric's 512 spp, both meshes: stream count (default 2), samples per batch
# (default 32), the shadow-ray kernel's LDS-staged top levels (default 0)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/knobs_final.txt
one() { python3 bench.py --mesh $1 --no-extras --spp 512 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
{ echo "# runtime knobs at 512 spp; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do echo "-- $mesh"
    echo "default            $(one $mesh)"
    for s in 1 3 4 5; do echo "MIRRES_STREAMS=$s   $(MIRRES_STREAMS=$s one $mesh)"; done
    for k in 16 64; do echo "MIRRES_PT_BATCH=$k $(MIRRES_PT_BATCH=$k one $mesh)"; done
    echo "MIRRES_TOPQ=85     $(MIRRES_TOPQ=85 one $mesh)"
    echo "default            $(one $mesh)"
  done
} 2>&1 | tee $O
