"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
streaming reads (MI355X_MICROARCH.md, HBM section), so the read side is doubled as that guide prescribes."""
import glob, sqlite3, sys
from collections import defaultdict
def load(d, counter):
    con = sqlite3.connect(glob.glob(d + "/*/*.db")[0]); cur = con.cursor()
    out = defaultdict(lambda: [0, 0.0])
    for name, val in cur.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
        k = name.replace("(anonymous namespace)::", "")[:70]
        out[k][0] += 1; out[k][1] += val
    return out
f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
print("%-72s %7s %14s %14s %14s" % ("kernel", "calls", "fetch MB/call*", "write MB/call", "total GB"))
tot = 0
for k in sorted(f, key=lambda k: -(2 * f[k][1] + w.get(k, [0, 0])[1])):
    c = f[k][0]; fb = 2 * f[k][1] * 1024; wb = w.get(k, [0, 0.0])[1] * 1024
    tot += fb + wb
    if fb + wb > 5e7:
        print("%-72s %7d %14.2f %14.2f %14.3f" % (k, c, fb / c / 1e6, wb / max(w.get(k, [1, 0])[0], 1) / 1e6, (fb + wb) / 1e9))
print("* FETCH_SIZE x2 (gfx950 correction).  total %.2f GB" % (tot / 1e9))
