import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
ops.set_activation_dtype(dt)
td = torch.bfloat16 if dt == "bf16" else torch.float32
for xs, cout, s in [((16, 128, 128, 48), 96, 2), ((16, 64, 64, 96), 192, 2)]:
    g = ops.ConvSpec((3, 3), stride=s).geom(xs, cout)
    x = torch.randn(xs, device="cuda").to(td)
    gy = torch.randn((xs[0], g.out_h, g.out_w, cout), device="cuda").to(td)
    for _ in range(6):
        ops.conv_wgrad(x, gy, g, (3, 3, xs[-1], cout))
torch.cuda.synchronize()
