#!/bin/bash
# round 6, run 3: straight-line child selection with integer keys (4), explicit lane-mask logic (5), + v_addc on the mask (6), all with the short refill arithmetic
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
O=gpurun_out/r06/ab_any_sel456.txt
{ echo "# k_trace_any4q variants, csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())'); microbenchmark scripts/dev_any_pmc.py 1600 7 10 mode 0 (10 launches), both meshes"
  for mesh in icosphere clustered; do
    echo "== $mesh"
    for v in base s1l s4l s5l s6l base s1l s4l s5l s6l; do
      if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
      echo "$v  $(MIRRES_MESH=$mesh timeout 300 python3 scripts/dev_any_pmc.py 1600 7 10 0 2>&1 | tail -1)"
    done
  done
  unset MIRRES_LIB
  echo "== frame hashes (8 spp, icosphere / clustered)"
  for v in base s6l; do
    if [ $v = base ]; then unset MIRRES_LIB; else export MIRRES_LIB=$PWD/ab/libmirres_$v.so; fi
    echo "$v  $(timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"
    echo "$v  $(MIRRES_MESH=clustered timeout 300 python3 scripts/dev_frame_hash.py 8 2>&1 | tail -1)"
  done
  unset MIRRES_LIB
  echo "== frames, 128 spp (bench.py --no-extras --steps 3 --warmup 1), interleaved twice"
  for mesh in icosphere clustered; do echo "-- $mesh"; MESH=$mesh SPP=128 bash scripts/dev_ab_frame.sh s1l s4l s5l s6l; done
} 2>&1 | tee $O
