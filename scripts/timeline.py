"""Concurrency analysis of a rocprofv3 kernel trace (rocpd sqlite) of the pipelined benchmark loop.
usage: timeline.py DB [last_ms]
Prints the table columns once, then over the last `last_ms` of the trace: per-queue busy time, the histogram of how many
queues run a kernel at the same moment, the GPU-fill estimate (sum over running kernels of min(1, workgroups / 256 CUs /
workgroups-per-CU guess)), and the kernels with the largest total time ranked with their average duration -- to be compared
with the serial trace (scripts/prof_summary.py) for the slowdown each kernel family suffers from its co-runners."""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
gx = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")]
wx = [c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")]
sel = "start, end, name" + (", " + qcol if qcol else ", 0") + (", " + gx[0] if gx else ", 0") + (", " + wx[0] if wx else ", 1")
rows = cur.execute("select %s from kernels order by start" % sel).fetchall()
t_end = rows[-1][1]
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 500e6
rows = [r for r in rows if r[0] >= t_end - window]
t0 = rows[0][0]
span = rows[-1][1] - t0
print("window %.1f ms, %d kernels" % (span / 1e6, len(rows)))
busy_q = defaultdict(float)
for s, e, n, q, g, w in rows:
    busy_q[q] += e - s
for q, b in sorted(busy_q.items(), key=lambda t: -t[1]):
    print("queue %s: busy %.1f ms (%.0f%%)" % (q, b / 1e6, 100 * b / span))
# sweep line over start / end events
ev = []
for i, (s, e, n, q, g, w) in enumerate(rows):
    wgs = (g // max(w, 1)) if g else 0
    ev.append((s, 1, i, wgs))
    ev.append((e, -1, i, wgs))
ev.sort()
hist = defaultdict(float)
fill_time = defaultdict(float)
active = 0
fill = 0.0
last = ev[0][0]
for t, d, i, wgs in ev:
    hist[active] += t - last
    fill_time[min(int(fill * 4), 8)] += t - last
    last = t
    active += d
    fill += d * min(1.0, wgs / 512.0)
print("concurrency histogram (kernels in flight -> ms):", {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
print("fill estimate (sum min(1, workgroups/512), quarter bins -> ms):", {k / 4: round(v / 1e6, 1) for k, v in sorted(fill_time.items())})
agg = defaultdict(lambda: [0, 0.0])
for s, e, n, q, g, w in rows:
    a = agg[n]
    a[0] += 1
    a[1] += e - s
tot = sum(a[1] for a in agg.values())
print("sum of kernel durations %.1f ms (%.2fx the window)" % (tot / 1e6, tot / span))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-80s %6d %9.2f ms %8.1f us" % (n[:80], c, t / 1e6, t / c / 1e3))
