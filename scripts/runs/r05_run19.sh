#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# ordered closest-hit kernel alone (scripts/dev_any_pmc.py 1600 7 10, mode 2), icosphere: base = 5-comparator sorting network, CLINS = sorted insertion (rounds 1-4)"; bash scripts/dev_ab.sh 2 CLINS
  echo "# the same, lego-like mesh"; MIRRES_MESH=clustered bash scripts/dev_ab.sh 2 CLINS
  echo "# frames, icosphere"; bash scripts/dev_ab_frame.sh CLINS
  echo "# frames, clustered"; MESH=clustered bash scripts/dev_ab_frame.sh CLINS
} > gpurun_out/r05/ab_closest_sortnet.txt 2>&1
cat gpurun_out/r05/ab_closest_sortnet.txt
timeout -k 5 1500 python3 -m pytest tests/test_gpu_bvh.py tests/test_gpu_clustered.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r05/gputests_run19.txt 2>&1; tail -3 gpurun_out/r05/gputests_run19.txt
