#!/bin/bash
# Dev (GPU box): frame rate against the runtime knobs of mirres_render (streams, traversal blocks per CU, samples per batch)
run() { echo "$1 $(python bench.py --no-cpu-baseline --no-roofline --steps 3 2>&1 | grep -o '"value": [0-9.]*')"; }
run default
for s in 2 4 5; do export MIRRES_STREAMS=$s; run "streams=$s"; done; unset MIRRES_STREAMS
for b in 4 5 7 8; do export MIRRES_TRACE_BLOCKS_PER_CU=$b; run "trace_blocks=$b"; done; unset MIRRES_TRACE_BLOCKS_PER_CU
for k in 16 64; do export MIRRES_PT_BATCH=$k; run "batch=$k"; done; unset MIRRES_PT_BATCH
run default
