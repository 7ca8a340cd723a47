"""Times of the VGG / ResNet edge kernels of the generator step: conv1_1 data gradient (64 -> 3), the k3 s2 pool backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1000 / n


g = ops.ConvSpec((3, 3)).geom((8, 256, 256, 3), 64)
gy = torch.randn(8, 256, 256, 64, device="cuda"); w = torch.randn(3, 3, 3, 64, device="cuda")
print("conv1_1 dgrad: %.1f us" % t(lambda: ops.conv_dgrad(gy, w, g)))
x = torch.relu(torch.randn(8, 128, 128, 64, device="cuda")); gp = torch.randn(8, 64, 64, 64, device="cuda")
print("maxpool k3 s2 p1 bwd: %.1f us" % t(lambda: ops.maxpool_bwd(x, gp, 3, 2, 1)))
x2 = torch.relu(torch.randn(8, 256, 256, 64, device="cuda")); gp2 = torch.randn(8, 128, 128, 64, device="cuda")
print("maxpool k2 s2 bwd: %.1f us" % t(lambda: ops.maxpool_bwd(x2, gp2, 2, 2, 0)))
y = torch.randn(8, 256, 256, 64, device="cuda")
print("act_bwd relu 134 MB: %.1f us" % t(lambda: ops.act_bwd(gy, y, 2, 0.0)))
