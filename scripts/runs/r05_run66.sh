#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05 gpurun_out/pf
{ echo "# time line of the metric's frame (1600 x 1600, 512 spp x 2 frames) from a rocprofv3 kernel trace (scripts/dev_strip_timeline.py): the per-sample chain on the caller's stream, what runs beside it; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for mesh in icosphere clustered; do
    echo "## $mesh"
    rm -rf gpurun_out/pf/kt; MIRRES_MESH=$mesh timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pf/kt -o kt -- python3 scripts/dev_strip_one.py 1 0 512 2 > gpurun_out/pf/log 2>&1
    grep "per sample" gpurun_out/pf/log
    python3 scripts/dev_strip_timeline.py "$(find gpurun_out/pf/kt -name '*kernel_trace.csv' | head -1)" 0.4
  done; } 2>&1 | tee gpurun_out/r05/frame_timeline.txt
rm -rf gpurun_out/pf
