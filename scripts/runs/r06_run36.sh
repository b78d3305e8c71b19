#!/bin/bash
# round 6, run 36: k_spatial_resolve<5, true> with the next sample's initial reservoir requested up front (one dependent level less before the fused temporal merge):
# hoist (172 registers: two waves per SIMD), hoistw3 (held at three waves: 168 registers + 32 B scratch) against the shipped kernel (hoist0), where the chain is the period
# and on the frame; bit-identical by construction (frame hashes)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_resolve_hoist.txt
{ echo "# k_spatial_resolve: NR load hoisted; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for v in hoist0 hoist hoistw3 hoist0 hoist hoistw3; do
    export MIRRES_LIB=$PWD/ab/libmirres_$v.so
    echo "$v  training step: $(timeout 300 python3 scripts/train_step_bench.py --steps 6 2>&1 | grep '^stage-1' | cut -c29-45)   strip 4 of 8, 256 spp: $(timeout 300 python3 scripts/dev_strip_one.py 8 4 256 3 2>&1 | tail -1 | cut -c1-120)"
  done
  for v in hoist0 hoistw3; do export MIRRES_LIB=$PWD/ab/libmirres_$v.so; echo "$v  $(timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"; done
  unset MIRRES_LIB
  for mesh in icosphere clustered; do echo "-- $mesh"; MESH=$mesh SPP=128 timeout 900 bash scripts/dev_ab_frame.sh hoist hoistw3; done
} 2>&1 | tee $O
