"""Dev: pipeline-level A/B of Python-level module constants on ONE box (alternating bench.py --timing-only runs).
    python scripts/dev/ab_py.py <reps> <dtype> "stmt A" "stmt B" ...      ("" = the defaults)
Each variant is a Python statement executed before bench.main(), e.g. "import confignet_amd.ops as o; o.WINO4_MIN_WGS = 128"."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--child":
    sys.path.insert(0, root)
    stmt = sys.argv[2]
    sys.argv = ["bench.py", "--timing-only", "--dtype", sys.argv[3], "--steps", "30"]
    import bench
    orig_setup = bench.setup

    def setup(*a, **k):          # (after bench's own device selection / imports, before the model is built)
        exec(stmt, {})
        return orig_setup(*a, **k)
    bench.setup = setup
    bench.main()
    sys.exit(0)
reps, dtype, variants = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
res = {v: [] for v in variants}
for _ in range(reps):
    for v in variants:
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v or "pass", dtype], capture_output=True, text=True)
        line = [l for l in pr.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(pr.stderr[-600:])
        val = json.loads(line[-1])["value"] if line else float("nan")
        res[v].append(val)
        print("%-70s %.1f" % (v[:70] or "(default)", val), flush=True)
for v in variants:
    xs = sorted(res[v])
    print("MEDIAN %-70s %.1f   (%s)" % (v[:70] or "(default)", xs[len(xs) // 2], " ".join("%.1f" % x for x in res[v])))
