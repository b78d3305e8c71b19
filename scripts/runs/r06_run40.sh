#!/bin/bash
# round 6, run 40: the default batch size is 64 now (render.hip): frame hashes at 64 / 32 per batch (70-spp frames: three batch shapes), the whole GPU suite, then the
# collection (fourth run: render.hip is in the chain / train source groups; the traversal snapshot stays)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r06
{ echo "default (64)  $(timeout 300 python3 scripts/dev_frame_hash.py 70 2>&1 | tail -1)"; echo "PT_BATCH=32    $(MIRRES_PT_BATCH=32 timeout 300 python3 scripts/dev_frame_hash.py 70 2>&1 | tail -1)"
  echo "default (64)  $(MIRRES_MESH=clustered timeout 300 python3 scripts/dev_frame_hash.py 70 2>&1 | tail -1)"; echo "PT_BATCH=32    $(MIRRES_MESH=clustered MIRRES_PT_BATCH=32 timeout 300 python3 scripts/dev_frame_hash.py 70 2>&1 | tail -1)"; } | tee gpurun_out/r06/batch64_frame_hashes.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06/gpu_suite_e.txt
timeout 3000 bash scripts/profile_r06.sh 2>&1 | tail -30
