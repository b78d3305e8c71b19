#!/bin/bash
# round 6, run 34: TIMING builds of k_spatial_resolve (wrong frames) that bound what a one-lane-per-(pixel, neighbour) layout could return: probe1 = the
# per-neighbour arithmetic removed (loads kept), probe2 = the neighbour gathers removed (arithmetic kept), probe3 = both. Measured where the chain IS the period (strip 4 of 8:
# kernel trace -> duration of the resolve and the chain's period; the 800^2 x 32 spp training step) and on the full 128-spp frame.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06 gpurun_out/probe
O=gpurun_out/r06/ab_resolve_probes.txt
{ echo "# k_spatial_resolve<5, true> timing probes; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())') + probes"
  for v in probe0 probe1 probe2 probe3 probe0 probe1 probe2 probe3; do
    export MIRRES_LIB=$PWD/ab/libmirres_$v.so
    D=$PWD/gpurun_out/probe/$v; rm -rf $D; mkdir -p $D
    timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 scripts/dev_strip_one.py 8 4 128 3 > $D/log 2>&1
    T=$(find $D -name '*kernel_trace.csv' | head -1)
    echo "== $v  strip 4 of 8: $(tail -1 $D/log | cut -c1-110)"
    python3 scripts/dev_strip_timeline.py $T 2>&1 | grep -i "resolve\|period\|spatial_gen\|any4q" | cut -c1-200
    echo "   $v training step: $(timeout 300 python3 scripts/train_step_bench.py --steps 6 2>&1 | grep '^stage-1' | cut -c29-45)"
  done
  unset MIRRES_LIB
  for mesh in icosphere; do echo "-- $mesh 128-spp frame"; MESH=$mesh SPP=128 timeout 900 bash scripts/dev_ab_frame.sh probe0 probe1 probe2 probe3; done
} 2>&1 | tee $O
