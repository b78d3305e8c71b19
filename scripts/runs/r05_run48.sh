#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05 gpurun_out/pf
O=gpurun_out/r05/strip_timeline.txt
{ echo "# strip 4 of 8 (icosphere, 1600 x 1600), 128 spp x 3 frames; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for b in 8 2 4 16; do echo "## MIRRES_TRACE_BLOCKS_PER_CU=$b"; MIRRES_TRACE_BLOCKS_PER_CU=$b timeout 300 python3 scripts/dev_strip_one.py 8 4 128 3 2>&1 | grep "per sample"; done
  echo "## whole frame, for scale"; timeout 300 python3 scripts/dev_strip_one.py 1 0 128 2 2>&1 | grep "per sample"
  echo "## kernel trace of the strip (timeline)"
  rm -rf gpurun_out/pf/kt; timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pf/kt -o kt -- python3 scripts/dev_strip_one.py 8 4 128 3 > gpurun_out/pf/log 2>&1
  grep "per sample" gpurun_out/pf/log
  python3 scripts/dev_strip_timeline.py "$(find gpurun_out/pf/kt -name '*kernel_trace.csv' | head -1)" 0.4
  echo "## kernel trace of the whole frame (timeline)"
  rm -rf gpurun_out/pf/kt; timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pf/kt -o kt -- python3 scripts/dev_strip_one.py 1 0 128 2 > gpurun_out/pf/log 2>&1
  grep "per sample" gpurun_out/pf/log
  python3 scripts/dev_strip_timeline.py "$(find gpurun_out/pf/kt -name '*kernel_trace.csv' | head -1)" 0.5
} 2>&1 | tee $O
rm -rf gpurun_out/pf
