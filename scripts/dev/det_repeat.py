"""Dev: is the deterministic mode bit-reproducible from process to process?  Runs tests/dp_run_helper.py N times (single-process
dispatch, then the forced-DP dispatch) and compares every array of the result files bit for bit."""
import os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
helper = os.path.join(root, "tests", "dp_run_helper.py")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ref = {}
for mode in ("single", "dp"):
    for i in range(n):
        env = {k: v for k, v in os.environ.items() if k != "CN_FORCE_DP"}
        if mode == "dp":
            env.update({"CN_FORCE_DP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29600 + i), "RANK": "0", "WORLD_SIZE": "1"})
        path = "/tmp/det_%s_%d.npz" % (mode, i)
        r = subprocess.run([sys.executable, helper, path, "0"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(mode, i, "FAILED rc", r.returncode, r.stderr[-1500:])
            continue
        d = np.load(path)
        base = ref.setdefault("single", d) if mode == "single" else ref["single"]
        diff = [(k, float(np.abs(d[k].astype(np.float64) - base[k].astype(np.float64)).max())) for k in d.files
                if k in base.files and d[k].shape == base[k].shape and not np.array_equal(d[k], base[k])]
        print(mode, i, "identical to the first single run" if not diff else "DIFFERS: %s" % diff[:6], flush=True)
