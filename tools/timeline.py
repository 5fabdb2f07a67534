#!/usr/bin/env python3
"""timeline.py <kernel_trace.csv> -- what the chip was doing during the headline loop: per kernel the mean duration, and for the
steady-state part of the trace the share of wall time in which 0, 1, 2, ... kernels were running, per-queue busy shares and the
mean period between consecutive k_scan launches (= the step time under the profiler)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
ev = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("sora::", "").strip()
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r["Queue_Id"]))
ev.sort()
# steady state: the last 60 % of k_scan launches
scans = [e for e in ev if e[2].startswith("k_scan")]
if len(scans) < 10:
    print("too few launches"); sys.exit(0)
t0 = scans[int(len(scans) * 0.4)][0]; t1 = scans[-3][0]
steady = [e for e in ev if e[0] >= t0 and e[1] <= t1]
per = collections.defaultdict(list)
for s, e, n, q in steady:
    per[n].append(e - s)
print("window %.3f ms, %d launches of k_scan -> period %.4f ms" % ((t1 - t0) / 1e6, sum(1 for e in steady if e[2].startswith("k_scan")),
      (t1 - t0) / 1e6 / max(1, sum(1 for e in scans if t0 <= e[0] < t1))))
for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("  %-28s n=%4d mean %.4f ms  sum %.3f ms (%.0f %% of the window)" % (n[:28], len(d), sum(d) / len(d) / 1e6, sum(d) / 1e6, 100.0 * sum(d) / (t1 - t0)))
pts = []
for s, e, n, q in steady:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
cur = 0; last = t0; hist = collections.Counter()
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
hist[cur] += t1 - last
tot = sum(hist.values())
print("  kernels running at once:", {k: "%.1f %%" % (100.0 * v / tot) for k, v in sorted(hist.items())})
# the same per kernel: how many launches of THAT kernel were running at once (time-weighted), and the mean
for name in sorted(per):
    p2 = []
    for s, e, n, q in steady:
        if n == name:
            p2.append((s, 1)); p2.append((e, -1))
    p2.sort(); cur = 0; last = t0; h = collections.Counter()
    for t, d in p2:
        h[cur] += t - last; last = t; cur += d
    h[cur] += t1 - last
    tt = sum(h.values())
    print("  %-28s at once: mean %.2f  %s" % (name[:28], sum(k * v for k, v in h.items()) / tt, {k: "%.0f %%" % (100.0 * v / tt) for k, v in sorted(h.items())}))
qs = collections.defaultdict(int)
for s, e, n, q in steady:
    qs[q] += e - s
print("  busy share per queue:", {q: "%.0f %%" % (100.0 * v / (t1 - t0)) for q, v in sorted(qs.items())})
# gaps inside one queue between consecutive kernels
byq = collections.defaultdict(list)
for s, e, n, q in steady:
    byq[q].append((s, e, n))
for q, L in sorted(byq.items()):
    L.sort(); gaps = collections.defaultdict(list)
    for a, b in zip(L, L[1:]):
        gaps[a[2][:12] + "->" + b[2][:12]].append(b[0] - a[1])
    print("  queue", q, {k: "%.1f us" % (sum(v) / len(v) / 1e3) for k, v in gaps.items()})
