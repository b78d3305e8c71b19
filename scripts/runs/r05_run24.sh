#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "# chunks per wave (MR_CHUNK_DIV; base = 4): shadow-ray kernel alone, icosphere"; bash scripts/dev_ab.sh 0 CD2 CD8 CD16
  echo "# ordered closest-hit kernel alone, icosphere"; bash scripts/dev_ab.sh 2 CD2 CD8 CD16
  echo "# frames, icosphere"; bash scripts/dev_ab_frame.sh CD8 CD16
  echo "# frames, clustered"; MESH=clustered bash scripts/dev_ab_frame.sh CD8 CD16; } > gpurun_out/r05/ab_chunk_div.txt 2>&1
cat gpurun_out/r05/ab_chunk_div.txt
