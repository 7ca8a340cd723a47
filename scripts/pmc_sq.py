"""Per-kernel SQ counters of a rocprofv3 --pmc run (rocpd sqlite): sums per kernel name and the ratios that
matter for an MFMA loop (quad-cycle counters; see MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import glob, sqlite3, sys
from collections import defaultdict
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
for name, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name.replace("(anonymous namespace)::", "")[:60]
    acc[k][cn] += val
    if cn == "SQ_WAVE_CYCLES": calls[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    print(k, "calls", calls[k])
    for cn, v in sorted(c.items()):
        print("   %-28s %14.0f  %6.1f%% of WAVE_CYCLES" % (cn, v, 100.0 * v / wc))
