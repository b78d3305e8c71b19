#!/bin/bash
# round 6, run 5: tests touched so far; counters + kernel trace of the training step's backward kernels; host / device cost of the per-sample halo exchange over RCCL
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests/test_gpu_rccl.py tests/test_gpu_render.py tests/test_gpu_adjoints.py tests/test_gpu_matnet.py tests/test_gpu_bvh.py tests/test_gpu_clustered.py -x -q 2>&1 | tail -15 | tee gpurun_out/r06/tests_run5.txt
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "round6 or band or training" 2>&1 | tail -5 | tee -a gpurun_out/r06/tests_run5.txt
rocprofv3 -L 2>/dev/null | grep -i "atomic" | head -20 > gpurun_out/r06/counters_atomic_list.txt
bash scripts/pmc_train.sh train > gpurun_out/r06/pmc_train.txt 2>&1; cp gpurun_out/pmc_train/summary.json gpurun_out/r06/pmc_train.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/pf; mkdir -p gpurun_out/pf
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o t -- python3 scripts/train_step_bench.py --steps 3 > gpurun_out/pf/log_tr 2>&1
find gpurun_out/pf/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06/train_step_kernel_stats_a.csv; grep '^stage-1' gpurun_out/pf/log_tr > gpurun_out/r06/train_step_a.txt
rm -rf gpurun_out/pf
for mesh in icosphere clustered; do MIRRES_MESH=$mesh timeout 600 python3 scripts/dev_halo_host_cost.py 512 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -8; done | tee gpurun_out/r06/halo_host_cost.txt
