"""Dev: where a strip's per-sample chain spends its time, from a rocprofv3 --kernel-trace CSV of scripts/dev_strip_one.py (start / end time stamp of every dispatch).
    python scripts/dev_strip_timeline.py <kernel_trace.csv> [skip_frames_fraction=0.25]
Prints, over the steady part of the run: per chain kernel (generate, pixel-pair shadow rays, resolve + next temporal) the mean duration and the mean gap to the chain kernel
before it; the chain's period per sample; how much of the wall clock some kernel of ANY stream was running, and how many kernels ran at once on average."""
import csv, sys, collections
path = sys.argv[1]; skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[0] >= lo]
def short(n):
    n = n.replace("mr::", "").replace("void ", "")
    return n.split("(")[0]
CH = ("k_spatial_gen", "k_trace_any4q<false, 0, 0, 1>", "k_spatial_resolve<5, true>", "k_temporal", "k_spatial_resolve<5, false>")
chain = [r for r in rows if short(r[2]).startswith(CH)]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
boundary = 0
for a, b in zip(chain, chain[1:]):
    g = max(0, b[0] - a[1])
    if g > 1e6: boundary += g          # frame boundary (LBVH rebuild, environment tables, denoiser, host): reported apart
    else: gap[short(b[2])].append(g)
for r in chain: dur[short(r[2])].append(r[1] - r[0])
n_samples = len([r for r in chain if short(r[2]).startswith("k_spatial_gen")])
span = chain[-1][1] - chain[0][0]
print("chain kernels of %d samples over %.2f ms: period %.1f us per sample" % (n_samples, span / 1e6, span / 1e3 / max(1, n_samples)))
tot_d = tot_g = 0
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d = sum(dur[k]) / 1e3 / n_samples; g = sum(gap[k]) / 1e3 / n_samples
    tot_d += d; tot_g += g
    print("  %-34s %5d launches  mean %7.1f us  (min %6.1f)  = %6.1f us per sample; gap before it: mean %5.1f us = %5.1f us per sample" %
          (k, len(dur[k]), sum(dur[k]) / 1e3 / len(dur[k]), min(dur[k]) / 1e3, d, sum(gap[k]) / 1e3 / max(1, len(gap[k])), g))
print("  chain kernels %.1f us + gaps %.1f us per sample; between frames %.2f ms in all = %.1f us per sample" % (tot_d, tot_g, boundary / 1e6, boundary / 1e3 / n_samples))
# device-wide: union of busy intervals and mean concurrency within the chain's span
ev = []
for s, e, *_ in rows:
    if e <= chain[0][0] or s >= chain[-1][1]: continue
    ev.append((max(s, chain[0][0]), 1)); ev.append((min(e, chain[-1][1]), -1))
ev.sort()
busy = 0; conc = 0; cur = 0; last = chain[0][0]
for t, d in ev:
    if cur > 0: busy += t - last
    conc += cur * (t - last); last = t; cur += d
print("some kernel running %.1f %% of that span; kernels in flight on average %.2f; other streams' kernel time per sample %.1f us" %
      (100.0 * busy / span, conc / span, (conc - sum(sum(v) for v in dur.values())) / 1e3 / n_samples))
oth = collections.defaultdict(lambda: [0, 0])
for s, e, n, *_ in rows:
    k = short(n)
    if not k.startswith(CH): oth[k][0] += e - s; oth[k][1] += 1
for k, (tt, c) in sorted(oth.items(), key=lambda kv: -kv[1][0])[:10]:
    print("    other: %-40s %5d launches, %7.1f us per sample" % (k[:40], c, tt / 1e3 / n_samples))

