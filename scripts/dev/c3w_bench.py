import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for xs, cout, s in [((16, 256, 256, 3), 48, 2), ((16, 256, 256, 3), 64, 1), ((8, 256, 256, 3), 48, 2)]:
    g = ops.ConvSpec((3, 3), stride=s).geom(xs, cout)
    x = torch.randn(xs, device="cuda")
    for dt in (torch.float32, torch.bfloat16):
        gy = torch.randn((xs[0], g.out_h, g.out_w, cout), device="cuda").to(dt)
        ops.C3_WGRAD = True
        a = t(lambda: ops.conv_wgrad(x, gy, g, (3, 3, 3, cout)))
        ops.C3_WGRAD = False
        ops.set_activation_dtype("bf16" if dt == torch.bfloat16 else "f32")
        b = t(lambda: ops.conv_wgrad(x, gy, g, (3, 3, 3, cout)))
        ops.set_activation_dtype("f32")
        mb = (gy.numel() * gy.element_size() + x.numel() * 4) / 1e6
        print("%s cout %d s%d gy %s: c3 kernel %.1f us (%.0f GB/s), generic %.1f us" % (xs, cout, s, dt, a, mb / a * 1e3, b))
