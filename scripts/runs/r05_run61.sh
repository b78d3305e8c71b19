#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
  timeout 300 python3 -c "import __graft_entry__ as G; G.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout 600 python3 bench.py 2>/dev/null | cut -c1-400; } | tee gpurun_out/r05/final_check.txt
