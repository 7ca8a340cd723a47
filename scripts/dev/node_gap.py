"""Cost of one dependent kernel node in a replayed graph (tiny kernels back to back on one stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
x = torch.zeros(256, device="cuda")
big = torch.zeros(8, 64, 64, 256, device="cuda")
s = torch.cuda.Stream()


def run(fn, inner=500, reps=20):
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


print("cn_zero of 1 KB      : %.2f us per node" % run(lambda: ops.zero_(x)))
print("torch add_ of 1 KB   : %.2f us per node" % run(lambda: x.add_(1.0)))
print("cn_zero of 33 MB     : %.2f us per node" % run(lambda: ops.zero_(big), inner=100))
a, b = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
s2 = torch.cuda.Stream()


def two():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        ops.zero_(b)
    ops.zero_(a)
    cur.wait_stream(s2)


print("fork + 2 nodes + join: %.2f us per pair" % run(two, inner=200))
