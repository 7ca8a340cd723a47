"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg, like --stats."""
import glob
import sqlite3
import sys

db = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/prof*/*/*.db")[-1]
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("# %s" % db)
print("# total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
adam = sum(r[1] for r in rows if "adam_kernel" in r[0])
if adam:
    # a second-stage iteration holds exactly 7 Adam launches (D, synth-D, latent-D: one network each; G: four networks)
    print("# %d adam_kernel launches = %.2f iterations' worth of step functions -> %.0f launches per iteration, %.2f ms of kernel time per iteration"
          % (adam, adam / 7.0, sum(r[1] for r in rows) / (adam / 7.0), tot / 1e6 / (adam / 7.0)))
print("%-90s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
for n, c, s, a, mn, mx in rows[:60]:
    print("%-90s %8d %12.3f %10.1f %10.1f %10.1f %6.2f" % (n[:90], c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
