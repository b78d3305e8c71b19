#!/bin/bash
# Dev: per-dispatch durations of the runtime's fill kernels (hipMemsetAsync) inside one bench frame
R=$PWD; export TMPDIR=/tmp MIRRES_STREAMS=${MIRRES_STREAMS:-1}
D=/tmp/ft; rm -rf $D; mkdir -p $D
(cd $D && timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 --spp 32 > $D/log 2>&1) || tail -5 $D/log
python3 - $D/t_kernel_trace.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
prev = None; hist = collections.Counter(); big = []
for i, r in enumerate(rows):
    if 'fillBuffer' in r['Kernel_Name']:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        hist[int(d // 10) * 10] += 1
        if d > 30: big.append((d, r.get('Grid_Size_X', r.get('Grid_Size', '?')), rows[i - 1]['Kernel_Name'][:40] if i else '', rows[i + 1]['Kernel_Name'][:40] if i + 1 < len(rows) else ''))
print(sorted(hist.items()))
for b in big[:30]: print("%8.1f us grid %s  after %s  before %s" % b)
PY
