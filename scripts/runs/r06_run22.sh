#!/bin/bash
# round 6, run 22: ordered closest-hit kernel with unconditional pushes (-DMR_CL_SEL=1): microbenchmark (mode 2), frame hashes, 128-spp frames
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_closest_sel.txt
{ echo "# k_trace_closest4 with unconditional pushes; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do for v in base clsel base clsel; do
      if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
      echo "$mesh $v  $(MIRRES_MESH=$mesh timeout 300 python3 scripts/dev_any_pmc.py 1600 7 10 2 2>&1 | tail -1)"
  done; done
  unset MIRRES_LIB
  for v in base clsel; do
    if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
    echo "$v  $(timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"; echo "$v  $(MIRRES_MESH=clustered timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"
  done
  unset MIRRES_LIB
  for mesh in icosphere clustered; do echo "-- $mesh"; MESH=$mesh SPP=128 bash scripts/dev_ab_frame.sh clsel; done
} 2>&1 | tee $O
