#!/bin/bash
# A/B of library builds: frame rate (bench.py, interleaved runs) and per-kernel averages (rocprofv3 --kernel-trace --stats of one serialised
# bench run each, MIRRES_STREAMS=1 so that kernel durations are not stretched by the other streams).
#   scripts/dev_ab_stats.sh "<kernel name regex>" <libA.so> [<libB.so> ...]      (the in-tree build is always variant 0)
R=$PWD; PAT=$1; shift
export TMPDIR=/tmp
for i in 1 2; do
  n=0
  for lib in "" "$@"; do
    if [ -n "$lib" ]; then export MIRRES_LIB=$lib; else unset MIRRES_LIB; fi
    echo "variant $n $(python bench.py --no-cpu-baseline --no-roofline --steps 3 2>&1 | grep -o '"value": [0-9.]*')"; n=$((n+1))
  done
done
n=0
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export MIRRES_LIB=$lib; else unset MIRRES_LIB; fi
  export MIRRES_STREAMS=1
  D=/tmp/ab_$n; rm -rf $D; mkdir -p $D
  (cd $D && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 --spp 32 > $D/log 2>&1) || tail -5 $D/log
  unset MIRRES_STREAMS
  echo "== variant $n (serialised, 32 spp)"; python3 - "$D/t_kernel_stats.csv" "$PAT" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']): print("%-50s %5s calls %9.1f us avg" % (r['Name'][:50], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  n=$((n+1))
done
