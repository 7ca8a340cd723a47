"""Run ONE convolution shape `reps` times (PMC / trace target):
   python scripts/conv_one.py fwd|dgrad|wgrad n h w cin cout k stride [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from confignet_amd import ops

kind = sys.argv[1]
n, h, w, cin, cout, k, st = map(int, sys.argv[2:9])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
spec = ops.ConvSpec((k, k), stride=st)
x = torch.randn(n, h, w, cin, device="cuda")
wt = torch.randn(k, k, cin, cout, device="cuda")
b = torch.randn(cout, device="cuda")
g = spec.geom(tuple(x.shape), cout)
gy = torch.randn(n, g.out_h, g.out_w, cout, device="cuda")
fn = {"fwd": lambda: ops.conv_fwd(x, wt, b, g, 1, 0.2), "dgrad": lambda: ops.conv_dgrad(gy, wt, g),
      "wgrad": lambda: ops.conv_wgrad(x, gy, g, tuple(wt.shape))}[kind]
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print("%s n%d %dx%d cin%d cout%d k%d s%d: %.1f us, %.1f TFLOP/s (dense count)" % (
    kind, n, h, w, cin, cout, k, st, us, 2.0 * n * g.out_h * g.out_w * k * k * cin * cout / us / 1e6))
