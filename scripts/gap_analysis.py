"""Idle-gap analysis of a rocprofv3 kernel trace (rocpd sqlite): union of kernel intervals over the last part of the
run, the largest gaps and the kernels that end before / start after each gap."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select start, end, name from kernels order by start").fetchall()
t_end = rows[-1][1]
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 400e6      # last N ms
rows = [r for r in rows if r[0] >= t_end - window]
busy, gaps, cur_end, last_name = 0, [], rows[0][0], ""
for s, e, n in rows:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end, last_name, n))
        busy += e - s
        cur_end, last_name = e, n
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end, last_name = e, n
span = rows[-1][1] - rows[0][0]
print("window %.1f ms, busy %.1f ms (%.1f%%), %d kernels, %d gaps" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rows), len(gaps)))
print("gap histogram: >1ms %d (%.1f ms), 100us-1ms %d (%.1f ms), <100us %d (%.1f ms)" % (
    sum(g[0] > 1e6 for g in gaps), sum(g[0] for g in gaps if g[0] > 1e6) / 1e6,
    sum(1e5 < g[0] <= 1e6 for g in gaps), sum(g[0] for g in gaps if 1e5 < g[0] <= 1e6) / 1e6,
    sum(g[0] <= 1e5 for g in gaps), sum(g[0] for g in gaps if g[0] <= 1e5) / 1e6))
for g in sorted(gaps, reverse=True)[:25]:
    print("%8.3f ms at t=%9.3f ms  after %-50s before %-50s" % (g[0] / 1e6, (g[1] - rows[0][0]) / 1e6, g[2][:50], g[3][:50]))
