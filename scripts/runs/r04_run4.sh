#!/bin/bash
# Round 4, GPU run 4: temporal merge fused into the spatial resolve (A/B by MIRRES_FUSE_TEMPORAL), strip overlap (tests + synthetic exchange), render / fullsize / clustered tests
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
( time timeout 1800 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_passes.py tests/test_gpu_rccl.py tests/test_gpu_fullsize.py tests/test_gpu_clustered.py tests/test_gpu_training.py -m gpu -q ) > gpurun_out/r04/gpu_tests_fuse.log 2>&1
tail -8 gpurun_out/r04/gpu_tests_fuse.log
out=gpurun_out/r04/ab_fuse_temporal.txt; : > $out
one() { python3 bench.py --mesh $1 --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do for f in 0 1; do echo "mesh $mesh fuse_temporal $f: $(MIRRES_FUSE_TEMPORAL=$f one $mesh)" >> $out; done; done; done
cat $out
python3 scripts/dev_strip_overlap.py 8 50 128 > gpurun_out/r04/strip_overlap.txt 2>&1; python3 scripts/dev_strip_overlap.py 2 50 128 >> gpurun_out/r04/strip_overlap.txt 2>&1
cat gpurun_out/r04/strip_overlap.txt
