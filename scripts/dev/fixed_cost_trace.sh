#!/bin/bash
# kernel durations (GPU side) of scripts/dev/fixed_cost.py from a rocprofv3 kernel trace
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/fc
rocprofv3 --kernel-trace -d /tmp/fc -- python $R/scripts/dev/fixed_cost.py > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3
db = glob.glob("/tmp/fc/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
t = "kernels" if "kernels" in tabs else [x for x in tabs if "kernel_dispatch" in x][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t).fetchall()]
gcol = [c for c in cols if "grid" in c.lower() and c.lower().endswith("x")]
gcol = gcol[0] if gcol else "0"
rows = cur.execute("select name, %s, avg(end-start), min(end-start), count(*) from %s group by name, %s order by name" % (gcol, t, gcol)).fetchall()
for n, gsz, a, mn, c in rows:
    print("%-60s grid %8d avg %7.1f us min %7.1f x %d" % (n[:60], gsz, a / 1e3, mn / 1e3, c))
PY
