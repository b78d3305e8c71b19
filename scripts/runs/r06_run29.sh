#!/bin/bash
# round 6, run 29: per-wave LDS leaf queue in the shadow-ray kernel (-DMR_ANY_LEAFQ=1) over its two thresholds (entries waiting / lanes with nothing
# left but queued leaves): microbenchmark with checksum on both meshes, frame hashes, 128-spp frames
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_leaf_queue.txt
V="lq32_8 lq48_8 lq16_4 lq32_16 lq64_24 lq24_6 lq48_16"
{ echo "# k_trace_any4q with a per-wave LDS leaf queue, variants lq<entries>_<blocked lanes>; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do for v in base $V base $V; do
      if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
      echo "$mesh $v  $(MIRRES_MESH=$mesh timeout 120 python3 scripts/dev_any_pmc.py 1600 7 10 0 2>&1 | tail -1)"
  done; done
  unset MIRRES_LIB
  for v in base lq32_8 lq16_4; do
    if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
    echo "$v  $(timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"; echo "$v  $(MIRRES_MESH=clustered timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"
  done
  unset MIRRES_LIB
  for mesh in icosphere clustered; do echo "-- $mesh"; MESH=$mesh SPP=128 timeout 900 bash scripts/dev_ab_frame.sh $V; done
} 2>&1 | tee $O
