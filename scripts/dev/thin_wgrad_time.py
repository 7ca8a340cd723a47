import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
x = torch.randn(8, 128, 128, 32, device="cuda"); spec = ops.ConvSpec((4, 4), up=1); g = spec.geom(tuple(x.shape), 3)
gy = torch.randn(8, 256, 256, 3, device="cuda")
for _ in range(3): ops.conv_wgrad(x, gy, g, (4, 4, 32, 3))
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(20): ops.conv_wgrad(x, gy, g, (4, 4, 32, 3))
e1.record(); torch.cuda.synchronize(); print("thin wgrad enabled=%s: %.1f us" % (ops.THIN_WGRAD, e0.elapsed_time(e1) * 50))
