#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05 gpurun_out/pf
O=gpurun_out/r05/strip_streams.txt
{ echo "# strip 4 of 8 (icosphere, 1600 x 1600), 512 spp x 2 frames, us per sample; csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())')"
  for st in 2 3 4 5; do for k in 32 64 16; do echo "## MIRRES_STREAMS=$st MIRRES_PT_BATCH=$k: $(MIRRES_STREAMS=$st MIRRES_PT_BATCH=$k timeout 300 python3 scripts/dev_strip_one.py 8 4 512 2 2>&1 | grep 'per sample')"; done; done
  for b in 2 4 16; do echo "## MIRRES_TRACE_BLOCKS_PER_CU=$b: $(MIRRES_TRACE_BLOCKS_PER_CU=$b timeout 300 python3 scripts/dev_strip_one.py 8 4 512 2 2>&1 | grep 'per sample')"; done
  echo "## MIRRES_BULK_PRIO=low: $(MIRRES_BULK_PRIO=low timeout 300 python3 scripts/dev_strip_one.py 8 4 512 2 2>&1 | grep 'per sample')"
  echo "## whole frame: $(timeout 300 python3 scripts/dev_strip_one.py 1 0 512 2 2>&1 | grep 'per sample')"
  echo "## lego-like, strip 4 of 8"
  for st in 2 3 4; do echo "## MIRRES_STREAMS=$st: $(MIRRES_MESH=clustered MIRRES_STREAMS=$st timeout 300 python3 scripts/dev_strip_one.py 8 4 512 2 2>&1 | grep 'per sample')"; done
  echo "## whole frame: $(MIRRES_MESH=clustered timeout 300 python3 scripts/dev_strip_one.py 1 0 512 2 2>&1 | grep 'per sample')"
  echo "## kernel trace of the strip (timeline), 512 spp x 2 frames"
  rm -rf gpurun_out/pf/kt; timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pf/kt -o kt -- python3 scripts/dev_strip_one.py 8 4 512 2 > gpurun_out/pf/log 2>&1
  grep "per sample" gpurun_out/pf/log
  python3 scripts/dev_strip_timeline.py "$(find gpurun_out/pf/kt -name '*kernel_trace.csv' | head -1)" 0.4
} 2>&1 | tee $O
rm -rf gpurun_out/pf
