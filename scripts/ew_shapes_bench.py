"""Collect every nc_reduce / nc_lin2 launch of one second-stage iteration (256x256, batch 16), time each distinct
(shape, operands, flags) in isolation (20 launches back to back in a replayed graph) and report achieved GB/s against the bytes
the launch must move."""
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, ".")
from confignet_amd import ConfigNet, SyntheticFaceDataset, ops, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs

calls = OrderedDict()
o_red, o_lin = ops.nc_reduce, ops.nc_lin2


def nc_reduce(x1, x2=None, want_sum=True, want_dot=True, flags=0, slope=0.0, per_channel=False, **kw):
    k = ("reduce", tuple(x1.shape), x2 is not None, want_sum, want_dot, flags, per_channel)
    calls[k] = calls.get(k, 0) + 1
    return o_red(x1, x2, want_sum, want_dot, flags, slope, per_channel, **kw)


def nc_lin2(shape, x1=None, a1=None, x2=None, a2=None, b=None, flags=0, slope=0.0, per_channel=False, a3=None, b3=None, **kw):
    k = ("lin2", tuple(shape), x1 is not None, x2 is not None, b is not None, flags, per_channel, a3 is not None)
    calls[k] = calls.get(k, 0) + 1
    return o_lin(shape, x1, a1, x2, a2, b, flags, slope, per_channel, a3, b3, **kw)


ops.nc_reduce, ops.nc_lin2 = nc_reduce, nc_lin2
import confignet_amd.functional as F
for mod in (F,):
    if hasattr(mod, "nc_reduce"):
        mod.nc_reduce = nc_reduce
    if hasattr(mod, "nc_lin2"):
        mod.nc_lin2 = nc_lin2
np.random.seed(0)
ds = SyntheticFaceDataset(64, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 16, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
m.setup_training(None, ds, 0, real_training_set=ds)
m.training_iteration(ds, ds, optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"]))
torch.cuda.synchronize()
ops.nc_reduce, ops.nc_lin2 = o_red, o_lin
del m
torch.cuda.empty_cache()


def timeit(fn, inner=20, reps=5):
    """Per-call time inside a replayed graph (no host dispatch in the figure: eager launches of these kernels are host-bound)."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
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


rows = []
for k, cnt in calls.items():
    shape = k[1]
    numel = int(np.prod(shape))
    n, c = shape[0], shape[-1]
    if k[0] == "reduce":
        _, _, has2, ws, wd, flags, pc = k
        x1 = torch.randn(shape, device="cuda")
        x2 = torch.randn(shape, device="cuda") if has2 else None
        us = timeit(lambda: o_red(x1, x2, ws, wd, flags, 0.3, pc))
        nbytes = 4 * numel * (2 if has2 else 1)
        desc = "reduce x2=%d sum=%d dot=%d flags=%d pc=%d" % (has2, ws, wd, flags, pc)
    else:
        _, _, h1, h2, hb, flags, pc, h3 = k
        x1 = torch.randn(shape, device="cuda") if h1 else None
        x2 = torch.randn(shape, device="cuda") if h2 else None
        cs = (c,) if pc else (n, c)
        a1 = torch.randn(cs, device="cuda") if h1 else None
        a2 = torch.randn(cs, device="cuda") if h2 else None
        b = torch.randn(cs, device="cuda") if hb else None
        a3 = torch.randn(cs, device="cuda") if h3 else None
        us = timeit(lambda: o_lin(shape, x1, a1, x2, a2, b, flags, 0.3, pc, a3, a3))
        nbytes = 4 * numel * (1 + int(h1) + int(h2))
        desc = "lin2 x1=%d x2=%d b=%d flags=%d pc=%d a3=%d" % (h1, h2, hb, flags, pc, h3)
    rows.append((us * cnt, cnt, us, nbytes / us / 1e3, nbytes / 1e6, shape, desc))
tot = sum(r[0] for r in rows)
print("total nc_reduce/nc_lin2 time per iteration (isolated): %.2f ms, %d distinct, %d launches" % (tot / 1e3, len(rows), sum(r[1] for r in rows)))
print("%4s %9s %8s %8s  %-24s %s" % ("cnt", "us/call", "GB/s", "MB", "shape", "op"))
for r in sorted(rows, reverse=True)[:70]:
    print("%4d %9.1f %8.0f %8.1f  %-24s %s  (%.1f%%)" % (r[1], r[2], r[3], r[4], str(r[5]), r[6], 100 * r[0] / tot))
