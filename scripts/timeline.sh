#!/bin/bash
# kernel timeline of one frame: per-queue busy time and how many queues are busy over the frame (which stream is the critical path?)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl gpurun_out/out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/tl/log 2>&1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/tl/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(fs[0])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ','').replace('mr::','')[:26], r.get('Queue_Id','?')) for r in rows]
ev.sort()
# the last frame = the last 45 % of the trace by time (warmup frame first); locate by the k_prep launches
preps = [e for e in ev if e[2].startswith('k_prep')]
t0 = preps[-1][0]; fr = [e for e in ev if e[0] >= t0]
t1 = max(e[1] for e in fr)
print("frame span %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(fr)))
byq = collections.defaultdict(list)
for e in fr: byq[e[3]].append(e)
for q, es in byq.items():
    busy = sum(e[1] - e[0] for e in es)
    names = collections.Counter(e[2] for e in es).most_common(3)
    print("queue %s: %5d kernels, busy %.1f ms, first %.1f last %.1f ms; %s" % (q, len(es), busy / 1e6, (min(e[0] for e in es) - t0) / 1e6, (max(e[1] for e in es) - t0) / 1e6, names))
# concurrency histogram
pts = []
for e in fr: pts.append((e[0], 1)); pts.append((e[1], -1))
pts.sort()
hist = collections.Counter(); cur = 0; last = t0
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
print("busy-queue histogram (ms):", {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
# time with exactly one kernel running: which kernels
solo = collections.Counter(); cur = []; last = t0
active = {}
pts2 = []
for i, e in enumerate(fr): pts2.append((e[0], 1, i)); pts2.append((e[1], -1, i))
pts2.sort()
act = set(); last = t0
for t, d, i in pts2:
    if len(act) == 1: solo[fr[next(iter(act))][2]] += t - last
    last = t
    if d == 1: act.add(i)
    else: act.discard(i)
print("solo time by kernel (ms):", [(k, round(v / 1e6, 1)) for k, v in solo.most_common(8)])
PY
rm -rf gpurun_out/tl
