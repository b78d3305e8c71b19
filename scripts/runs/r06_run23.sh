#!/bin/bash
# round 6, run 23: shadow-ray kernel with the pop as straight-line code (-DMR_ANY_POP=1): microbenchmark, frame hashes, 128-spp frames
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_any_pop.txt
{ echo "# k_trace_any4q with a straight-line pop; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do for v in base pop base pop; do
      if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
      echo "$mesh $v  $(MIRRES_MESH=$mesh timeout 300 python3 scripts/dev_any_pmc.py 1600 7 10 0 2>&1 | tail -1)"
  done; done
  unset MIRRES_LIB
  for v in base pop; do
    if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
    echo "$v  $(timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"; echo "$v  $(MIRRES_MESH=clustered timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"
  done
  unset MIRRES_LIB
  for mesh in icosphere clustered; do echo "-- $mesh"; MESH=$mesh SPP=128 bash scripts/dev_ab_frame.sh pop; done
} 2>&1 | tee $O
