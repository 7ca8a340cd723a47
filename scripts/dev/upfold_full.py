import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops, functional as F
cases = [((16, 4, 4, 4, 512), (3, 3, 3), 256), ((16, 8, 8, 8, 256), (3, 3, 3), 128), ((16, 16, 16, 256), (4, 4), 64),
         ((16, 32, 32, 64), (4, 4), 32), ((16, 64, 64, 32), (4, 4), 32), ((8, 8, 8, 8, 256), (3, 3, 3), 128)]
for dtype in sys.argv[1:] or ["f32"]:
    ops.set_activation_dtype(dtype)
    for xs, k, cout in cases:
        x = torch.randn(xs, device="cuda")
        if dtype == "bf16":
            x = x.to(torch.bfloat16)
        x.requires_grad_(True)
        w = (torch.randn(*k, xs[-1], cout, device="cuda") / math.sqrt(np.prod(k) * xs[-1])).requires_grad_(True)
        b = torch.randn(cout, device="cuda").requires_grad_(True)
        spec = ops.ConvSpec(k, up=1)
        print(dtype, xs, k, cout, "fwd", flush=True)
        y = F.conv(x, w, b, spec)
        torch.cuda.synchronize()
        print("  bwd", flush=True)
        g = torch.autograd.grad(y.float().sum(), [x, w, b])
        torch.cuda.synchronize()
        # compare with the un-collapsed path
        ops.UPFOLD = False
        y2 = F.conv(x, w, b, spec)
        g2 = torch.autograd.grad(y2.float().sum(), [x, w, b])
        ops.UPFOLD = True
        torch.cuda.synchronize()
        print("  max diff fwd %.3e  gx %.3e (scale %.2e)  gw %.3e (scale %.2e)" % (float((y.float() - y2.float()).abs().max()),
              float((g[0].float() - g2[0].float()).abs().max()), float(g2[0].float().abs().max()),
              float((g[1] - g2[1]).abs().max()), float(g2[1].abs().max())), flush=True)
