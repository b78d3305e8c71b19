#!/bin/bash
# round 6, run 32: which passing child the shadow ray descends into first (-DMR_ANY_ORDER: 0 nearest, 1 largest exit distance, 2 longest chord; l20 = 20 stack entries in LDS):
# microbenchmark with checksum on both meshes, frame hashes, counting-kernel statistics, 128-spp frames
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_any_order.txt
V="ord1 ord2 ord1l20 ord2l20"
{ echo "# k_trace_any4q child order; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do for v in base $V base $V; do
      if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
      echo "$mesh $v  $(MIRRES_MESH=$mesh timeout 120 python3 scripts/dev_any_pmc.py 1600 7 10 0 2>&1 | tail -1)"
  done; done
  unset MIRRES_LIB
  for v in base ord1 ord2; do
    if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
    echo "$v  $(timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"; echo "$v  $(MIRRES_MESH=clustered timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"
    echo "## $v"; timeout 300 python3 scripts/dev_leaf_branch.py 4 2>&1 | grep -v amdgpu.ids | tail -2
  done
  unset MIRRES_LIB
  for mesh in icosphere clustered; do echo "-- $mesh"; MESH=$mesh SPP=128 timeout 900 bash scripts/dev_ab_frame.sh $V; done
} 2>&1 | tee $O
