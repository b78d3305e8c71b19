#!/bin/bash
# Round 4, GPU run 3: private steering hierarchy A/B (MIRRES_PRIVATE_TREE=0: collapsed reference LBVH, 1: extended Morton codes) on both meshes — shadow-ray
# microbenchmark, frame at 128 spp — then the bvh / clustered / fullsize tests with the new tree
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_private_tree.txt; : > $out
for mesh in icosphere clustered; do
  for rep in 1 2; do
    for pt in 0 1; do
      echo "mesh $mesh private_tree $pt: $(MIRRES_MESH=$mesh MIRRES_PRIVATE_TREE=$pt python3 scripts/dev_any_pmc.py 1600 7 10 0 2>&1 | tail -1)" >> $out
    done
  done
  for pt in 0 1; do
    echo "mesh $mesh private_tree $pt frame: $(MIRRES_PRIVATE_TREE=$pt python3 bench.py --mesh $mesh --spp 128 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; any launch', r['launch_ms'], 'ms', r['grays_per_s'], 'Grays/s per_ray', r['per_ray']['any_production'], 'closest', r['closest']['launch_ms'], 'ms redo', r['closest']['redo_frac'], 'per_ray', r['per_ray']['closest'])")" >> $out
  done
done
cat $out
( time timeout 1500 python3 -m pytest tests/test_gpu_bvh.py tests/test_gpu_clustered.py tests/test_gpu_fullsize.py -m gpu -q ) > gpurun_out/r04/gpu_tests_tree.log 2>&1
tail -8 gpurun_out/r04/gpu_tests_tree.log
