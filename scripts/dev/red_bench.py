"""nc_reduce / nc_reduce4 on the iteration's shapes, timed inside a captured graph (no host dispatch in the figure):
    python scripts/dev/red_bench.py            # CN_RED_BLOCKS=... to sweep the workgroup target"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops

SHAPES = [((80, 128, 128, 48), True), ((64, 64, 64, 96), True), ((16, 128, 128, 48), True), ((8, 256, 256, 64), True),
          ((8, 256, 256, 64), False), ((16, 64, 64, 96), True), ((48, 32, 32, 192), True), ((16, 32, 32, 192), True),
          ((8, 64, 64, 256), False), ((16, 16, 16, 384), True), ((16, 8, 8, 512), True), ((8, 16, 16, 16, 128), True)]


def graph_time(fn, inner=20, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (inner * reps)


print("%-26s %-8s %9s %8s" % ("shape", "op", "us/call", "GB/s"))
for shape, two in SHAPES:
    x1 = torch.randn(shape, device="cuda")
    x2 = torch.randn(shape, device="cuda") if two else None
    numel = x1.numel()
    us = graph_time(lambda: ops.nc_reduce(x1, x2, True, True, 2 if two else 0, 0.3))
    print("%-26s %-8s %9.1f %8.0f" % (shape, "reduce" + ("2" if two else "1"), us, 4 * numel * (2 if two else 1) / us / 1e3))
    if len(shape) == 4 and not two or shape[0] in (80, 64, 48):
        us = graph_time(lambda: ops.nc_reduce4(x1, 0.3))
        print("%-26s %-8s %9.1f %8.0f" % (shape, "reduce4", us, 4 * numel / us / 1e3))
    del x1, x2
