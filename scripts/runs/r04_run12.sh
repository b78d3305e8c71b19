#!/bin/bash
# Round 4, GPU run 12: SAH top over prefix clusters (MIRRES_PRIVATE_TREE=2) against the extended-Morton tree (1): traversal tests with it, microbenchmark, frames, build time
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time MIRRES_PRIVATE_TREE=2 timeout 1500 python3 -m pytest tests/test_gpu_bvh.py tests/test_gpu_clustered.py -m gpu -q ) > gpurun_out/r04/gpu_tests_sah.log 2>&1
tail -8 gpurun_out/r04/gpu_tests_sah.log | cut -c1-300
out=gpurun_out/r04/ab_sah_top.txt; : > $out
for mesh in icosphere clustered; do
  for rep in 1 2; do
    for pt in 1 2; do
      echo "mesh $mesh private_tree $pt: $(MIRRES_MESH=$mesh MIRRES_PRIVATE_TREE=$pt python3 scripts/dev_any_pmc.py 1600 7 10 0 2>&1 | tail -1)" >> $out
    done
  done
  for pt in 1 2; do
    echo "mesh $mesh private_tree $pt frame: $(MIRRES_PRIVATE_TREE=$pt python3 bench.py --mesh $mesh --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms; any launch', r['launch_ms'], 'ms', r['grays_per_s'], 'Grays/s per_ray', r['per_ray']['any_production'], 'closest', r['closest']['launch_ms'], 'ms per_ray', r['per_ray']['closest'], 'stack', r['private_stack_deepest'])")" >> $out
  done
done
for mesh in icosphere clustered; do for pt in 1 2; do echo "build time $mesh private_tree $pt: $(MIRRES_MESH=$mesh MIRRES_PRIVATE_TREE=$pt python3 scripts/dev_build_time.py 2>&1 | tail -2 | tr "\n" " ")" >> $out; done; done
cat $out
