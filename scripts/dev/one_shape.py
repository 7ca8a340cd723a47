"""One convolution shape launched repeatedly (for rocprofv3 kernel traces / PMC passes of a single kernel configuration).
python scripts/dev/one_shape.py kind n h w cin cout k stride loop ns cfg splits [reps]"""
import sys
import torch
import os as _os
NP = int(_os.environ.get("SWEEP_NP", "-1"))
KBS = int(_os.environ.get("SWEEP_KB", "0"))
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
from confignet_amd._lib import lib
kind = sys.argv[1]
n, h, w, cin, cout, k, st, loop, ns, cfg, splits = (int(a) for a in sys.argv[2:13])
reps = int(sys.argv[13]) if len(sys.argv) > 13 else 20
spec = ops.ConvSpec((k, k), stride=st)
x = torch.randn(n, h, w, cin, device="cuda")
wt = torch.randn(k, k, cin, cout, device="cuda")
b = torch.randn(cout, device="cuda")
g = spec.geom(tuple(x.shape), cout)
gy = torch.randn(n, g.out_h, g.out_w, cout, device="cuda")
ops.check(lib.cn_conv_loop_select(loop, KBS, ns, NP), "select")
ops.check(lib.cn_conv_tune(cfg, splits, 0), "tune")
fn = (lambda: ops.conv_fwd(x, wt, b, g, 1, 0.3)) if kind == "fwd" else (lambda: ops.conv_dgrad(gy, wt, g))
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record()
torch.cuda.synchronize()
print("%s us per call" % (e0.elapsed_time(e1) * 1e3 / reps))
