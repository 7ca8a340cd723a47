"""Per-kernel sums of every PMC counter in a rocprofv3 database (plus kernel durations): python scripts/dev/pmc_dump.py <dir> [name-substring]"""
import glob, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + "/*/*.db") + glob.glob(sys.argv[1] + "/*.db"))[0]
want = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
def tab(prefix):
    return next(t for t in tabs if t.startswith(prefix))
kd, ks, pmc, info = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
rows = c.execute("select d.id, s.kernel_name, d.end - d.start from %s d join %s s on d.kernel_id = s.id" % (kd, ks)).fetchall()
names = {r[0]: (r[1], r[2]) for r in rows}
cols = [r[1] for r in c.execute("pragma table_info(%s)" % pmc)]
ev = c.execute("select e.event_id, i.name, e.value from %s e join %s i on e.pmc_id = i.id" % (pmc, info)).fetchall() if "event_id" in cols else []
# event_id -> dispatch: rocpd_kernel_dispatch.event_id
d_ev = dict(c.execute("select event_id, id from %s" % kd).fetchall())
agg = {}
for eid, cname, val in ev:
    did = d_ev.get(eid)
    if did is None:
        continue
    kname, dur = names[did]
    if want not in kname:
        continue
    a = agg.setdefault(kname[:100], {"n": set(), "dur": {}, "c": {}})
    a["n"].add(did)
    a["dur"][did] = dur
    a["c"][cname] = a["c"].get(cname, 0.0) + val
for k, a in agg.items():
    n = len(a["n"])
    print("%s\n   launches %d  avg duration %.1f us" % (k, n, sum(a["dur"].values()) / n / 1e3))
    for cn, v in sorted(a["c"].items()):
        print("   %-28s %.4g per launch" % (cn, v / n))
