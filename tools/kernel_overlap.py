"""Device timeline of a rocprofv3 --kernel-trace CSV: how much of the wall time has 0, 1, 2+ kernels running, and the
mean duration of every kernel, over the kernels number lo*N .. hi*N of the trace in start order (pick the timed
region of bench.py that way)."""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
rows.sort()
rows = rows[int(lo * len(rows)):int(hi * len(rows))]
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], defaultdict(int)
for t, dlt in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += dlt
total = sum(hist.values())
print("window %.2f ms, kernels %d" % (total / 1e6, len(rows)))
for k in sorted(hist):
    print("  %d%s kernels running: %5.1f %%" % (k, "+" if k == 3 else "", 100.0 * hist[k] / total))
acc = defaultdict(lambda: [0, 0])
for s, e, n, q in rows:
    acc[n][0] += e - s; acc[n][1] += 1
for n, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("  %-50s calls %5d  mean %7.1f us  share of window %5.1f %%" % (n[:50], c, t / c / 1e3, 100.0 * t / total))

# per queue: share of the window with a kernel of that queue running, and where its idle time sits (by the kernel that
# ended before the gap and the one that started after it)
byq = defaultdict(list)
for s, e, n, q in rows:
    byq[q].append((s, e, n))
for q, ks in sorted(byq.items()):
    ks.sort()
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        if s1 > e0:
            g = gaps[(n0.split("::")[-1][:22], n1.split("::")[-1][:22])]
            g[0] += s1 - e0; g[1] += 1
    print("queue %s: %d kernels, busy %.1f %% of its span" % (q, len(ks), 100.0 * busy / max(span, 1)))
    for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:8]:
        print("    gap %-22s -> %-22s  %5d times  mean %6.1f us  %5.1f %% of span" % (a, b, c, t / c / 1e3, 100.0 * t / max(span, 1)))
