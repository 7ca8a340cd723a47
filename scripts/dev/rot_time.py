import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
grid = torch.randn(8, 16, 16, 16, 128, device="cuda"); gout = torch.randn_like(grid)
rot = torch.eye(3, device="cuda").repeat(8, 1, 1).contiguous() + 0.05 * torch.randn(8, 3, 3, device="cuda")
for need in (True, False):
    for _ in range(3): ops.rotate3d_bwd(grid, rot, gout, need)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): ops.rotate3d_bwd(grid, rot, gout, need)
    e1.record(); torch.cuda.synchronize(); print("rotate3d_bwd need_rot=%s: %.1f us" % (need, e0.elapsed_time(e1) * 50))
