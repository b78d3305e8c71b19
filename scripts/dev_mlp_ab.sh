#!/bin/bash
# Dev (GPU box): fp32-MFMA material MLP (default, bit-equal to the fmaf chain) against the split-f16 form (MIRRES_MLP_F16SPLIT=1): GEMM phase alone and the frame.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
for v in 0 1; do
  echo "== MIRRES_MLP_F16SPLIT=$v"
  MIRRES_MLP_F16SPLIT=$v python3 scripts/dev_mlp_bench.py
  MIRRES_MLP_F16SPLIT=$v python3 bench.py --spp 128 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frame', d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"
done
