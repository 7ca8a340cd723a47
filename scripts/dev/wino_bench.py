import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
shapes = [((8, 256, 256, 64), 64), ((8, 128, 128, 128), 128), ((8, 64, 64, 256), 256), ((8, 32, 32, 512), 512), ((8, 128, 128, 64), 128), ((16, 64, 64, 256), 256)]
def t(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
if os.environ.get('WINO_SHAPE'): shapes = [shapes[int(os.environ['WINO_SHAPE'])]]
for xs, cout in shapes:
    x = torch.randn(xs, device="cuda") * float(os.environ.get("WINO_XSCALE", "1")); w = torch.randn(3, 3, xs[-1], cout, device="cuda") * 0.05; b = torch.randn(cout, device="cuda")
    g = ops.ConvSpec((3, 3)).geom(xs, cout)
    fl = 2.0 * xs[0] * xs[1] * xs[2] * 9 * xs[-1] * cout
    ops.WINOGRAD = False
    td = t(lambda: ops.conv_fwd(x, w, b, g, 2, 0.0))
    ops.WINOGRAD = True
    ops.WINO4 = False
    tw = t(lambda: ops.conv_fwd(x, w, b, g, 2, 0.0))
    ops.WINO4 = True
    t4 = t(lambda: ops.conv_fwd(x, w, b, g, 2, 0.0)) if ops._wino4_ok(g, xs[-1], cout) else float("nan")
    print("F(4x4) %7.1f us (%5.1f TF-equivalent, MFMA %5.1f TF)   " % (t4, fl / t4 / 1e6, fl / 4 / t4 / 1e6), end="")
    print("%-22s cout %-4d direct %7.1f us (%5.1f TF)  winograd %7.1f us (%5.1f TF-equivalent, MFMA %5.1f TF)" % (xs, cout, td, fl / td / 1e6, tw, fl / tw / 1e6, fl * 4 / 9 / tw / 1e6))
