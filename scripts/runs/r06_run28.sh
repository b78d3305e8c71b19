#!/bin/bash
# round 6, run 28: the shadow-ray kernel's refill threshold (MR_REFILL: a wave fetches new rays when fewer than this many lanes hold one; 40 since round 3) re-measured on
# the round-6 kernel, whose refill is cheaper (short division / square root) and whose iterations no longer diverge in the selection and the pop: 32 / 48 / 56 against 40
# (the ordered closest-hit kernel held at 40): microbenchmark on both meshes, 128-spp frames
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_any_refill.txt
{ echo "# k_trace_any4q refill threshold; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do for v in base rf32 rf48 rf56 base rf32 rf48 rf56; do
      if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
      echo "$mesh $v  $(MIRRES_MESH=$mesh timeout 300 python3 scripts/dev_any_pmc.py 1600 7 10 0 2>&1 | tail -1)"
  done; done
  unset MIRRES_LIB
  for mesh in icosphere clustered; do echo "-- $mesh"; MESH=$mesh SPP=128 bash scripts/dev_ab_frame.sh rf32 rf48 rf56; done
} 2>&1 | tee $O
