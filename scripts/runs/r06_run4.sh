#!/bin/bash
# round 6, run 4: the band pipeline of the chain — parity tests, then what it does to the training step (configs[2]), an 800 x 800 frame and the full frame
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "band_pipeline or round6" 2>&1 | tail -15 | tee gpurun_out/r06/band_tests.txt
O=gpurun_out/r06/ab_bands.txt
{ echo "# band pipeline of the temporal->spatial chain: MIRRES_BANDS x MIRRES_CHAIN_STREAMS; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  echo "== stage-1 training step (scripts/train_step_bench.py --steps 6: 800 x 800, 32 spp)"
  for cfg in "1 1" "2 2" "4 2" "6 2" "8 2" "4 3" "6 3" "8 3" "12 3" "1 1"; do set -- $cfg
    echo "bands $1 streams $2: $(MIRRES_BANDS=$1 MIRRES_CHAIN_STREAMS=$2 timeout 300 python3 scripts/train_step_bench.py --steps 6 2>&1 | grep '^stage-1' | cut -c1-160)"
  done
  echo "== 800 x 800 internal frame, 128 spp (bench.py --res 400 --ssaa 2 --spp 128 --steps 3 --warmup 1 --no-extras)"
  for cfg in "1 1" "4 2" "6 2" "6 3" "8 3" "1 1"; do set -- $cfg
    echo "bands $1 streams $2: $(MIRRES_BANDS=$1 MIRRES_CHAIN_STREAMS=$2 timeout 300 python3 bench.py --res 400 --ssaa 2 --spp 128 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms", d["value"], "Msamples/s")')"
  done
  echo "== full frame 1600 x 1600, 128 spp, both meshes"
  for mesh in icosphere clustered; do for cfg in "1 1" "4 2" "8 3" "1 1"; do set -- $cfg
    echo "$mesh bands $1 streams $2: $(MIRRES_BANDS=$1 MIRRES_CHAIN_STREAMS=$2 timeout 300 python3 bench.py --mesh $mesh --spp 128 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], "ms", d["value"], "Msamples/s")')"
  done; done
} 2>&1 | tee $O
