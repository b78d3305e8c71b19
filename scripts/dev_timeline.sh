#!/bin/bash
# Dev (GPU box): kernel trace (start / end / queue of every dispatch) of one bench frame -> gpurun_out/timeline.csv
R=$PWD; export TMPDIR=/tmp
D=/tmp/tl; rm -rf $D; mkdir -p $D
(cd $D && timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 "$@" > $D/log 2>&1) || tail -5 $D/log
mkdir -p $R/gpurun_out; cp $D/t_kernel_trace.csv $R/gpurun_out/timeline.csv; head -2 $R/gpurun_out/timeline.csv | cut -c1-600; wc -l $R/gpurun_out/timeline.csv; grep -o '"value": [0-9.]*' $D/log
