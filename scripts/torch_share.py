"""Share of the GPU kernel time spent in PyTorch's own kernels (at::native::*, autograd accumulation adds, cat, fills)
in a rocprofv3 kernel trace (rocpd sqlite) of bench.py --serial.  usage: torch_share.py <trace_dir> <out.json>"""
import glob, json, os, sqlite3, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernels_hash

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tot = torch_t = 0.0
n = n_torch = 0
top = {}
adam = 0
for name, c, t in cur.execute("select name, count(*), sum(end-start) from kernels group by name"):
    tot += t
    n += c
    if "adam_kernel" in name:
        adam += c
    if "at::native" in name or name.startswith("void at::") or "c10::" in name:
        torch_t += t
        n_torch += c
        top[name[:100]] = t
out = {"kernels_hash": kernels_hash(), "torch_kernel_time_share": torch_t / tot, "torch_launch_share": n_torch / n,
       "launches": n, "torch_launches": n_torch, "adam_launches": adam,
       "launches_per_iteration": (round(n / (adam / 7.0), 1) if adam else None),
       "launches_per_iteration_how": "dispatches of the trace / (adam_kernel launches / 7): a second-stage iteration holds exactly 7 Adam launches",
       "top_torch_kernels_ms": {k: round(v / 1e6, 3) for k, v in sorted(top.items(), key=lambda kv: -kv[1])[:6]},
       "how": "rocprofv3 --kernel-trace on `python bench.py --serial --no-cpu-baseline --steps 10` (whole process: warm-up, "
              "capture-free eager iterations); share of summed kernel durations"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
