#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_training.py -m gpu -x -q 2>&1 | tail -3
{ echo "# short first batches (K/8, K/4, K/2, then K = 32; MIRRES_BATCH_RAMP=0: all K): ms per frame, bench.py --no-extras --no-cpu-baseline --no-roofline, 3 + 1 frames; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for spp in 32 64 128 512; do for r in 0 1 0 1; do
    echo "spp $spp, ramp $r: $(MIRRES_BATCH_RAMP=$r timeout 300 python3 bench.py --spp $spp --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms", d["value"], "Msamples/s")')"
  done; done
  for r in 0 1 0 1; do echo "lego-like, spp 512, ramp $r: $(MIRRES_BATCH_RAMP=$r timeout 300 python3 bench.py --mesh clustered --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms", d["value"], "Msamples/s")')"; done
  for r in 0 1 0 1; do echo "training step, ramp $r: $(MIRRES_BATCH_RAMP=$r timeout 300 python3 scripts/train_step_bench.py --steps 5 2>&1 | grep '^stage-1' | cut -c1-70)"; done
} 2>&1 | tee gpurun_out/r05/batch_ramp.txt
