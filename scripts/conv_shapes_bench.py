"""Collect every distinct implicit-GEMM launch of one second-stage iteration (256x256, batch 16) and time
each shape in isolation: per-shape TFLOP/s and share of the conv time.  Run on the GPU box."""
import ctypes
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, ".")
from confignet_amd import ConfigNet, SyntheticFaceDataset, ops, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs

FIELDS = [f[0] for f in ops.CnConvGeom._fields_]
calls = OrderedDict()
orig = {k: getattr(ops, k) for k in ("conv_fwd", "conv_dgrad", "conv_wgrad")}


def key(kind, g):
    return (kind,) + tuple(getattr(g, f) for f in FIELDS)


def rec(kind, g):
    calls[key(kind, g)] = calls.get(key(kind, g), 0) + 1


def conv_fwd(x, w, bias, g, act=0, slope=0.0):
    rec("fwd", g)
    return orig["conv_fwd"](x, w, bias, g, act, slope)


def conv_dgrad(gy, wt, g):
    rec("dgrad", g)
    return orig["conv_dgrad"](gy, wt, g)


def conv_wgrad(x, gy, g, ws, out=None, **kw):
    rec("wgrad", g)
    return orig["conv_wgrad"](x, gy, g, ws, out=out, **kw)


ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad = conv_fwd, conv_dgrad, conv_wgrad
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
DTYPE = sys.argv[2] if len(sys.argv) > 2 else "f32"          # "bf16": activation tensors with more than 4 channels in bf16
ops.set_activation_dtype(DTYPE)
np.random.seed(0)
ds = SyntheticFaceDataset(64, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": B, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
m.setup_training(None, ds, 0, real_training_set=ds)
m.training_iteration(ds, ds, optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"]))
torch.cuda.synchronize()
ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad = orig["conv_fwd"], orig["conv_dgrad"], orig["conv_wgrad"]
del m
torch.cuda.empty_cache()


def valid_pairs(out, k, s, dl, p, inn, up):
    c = 0
    for o in range(out):
        for kk in range(k):
            v = o * s - p + kk
            if v < 0 or v % dl or v // dl >= (inn << up):
                continue
            c += 1
    return c


rows = []
ONLY = os.environ.get("CONV_SHAPES_KIND")          # e.g. "wgrad": time only that kind (A/B runs)
for kk, cnt in calls.items():
    kind = kk[0]
    if ONLY and kind != ONLY:
        continue
    g = ops.CnConvGeom(*kk[1:])
    if kind == "dgrad":
        # effective geometry the kernel sees
        flops_g = ops.CnConvGeom(*kk[1:])
        e = dict(in_d=g.out_d, in_h=g.out_h, in_w=g.out_w, cin=g.cout, out_d=(g.in_d << g.up) if g.nd == 3 else 1,
                 out_h=g.in_h << g.up, out_w=g.in_w << g.up, cout=g.cin, s_d=1, s_h=1, s_w=1, dl_d=g.s_d, dl_h=g.s_h,
                 dl_w=g.s_w, p_d=g.k_d - 1 - g.p_d, p_h=g.k_h - 1 - g.p_h, p_w=g.k_w - 1 - g.p_w, up=0)
        for a, b in e.items():
            setattr(flops_g, a, b)
    else:
        flops_g = g
    f = flops_g
    flops = 2.0 * f.n * valid_pairs(f.out_d, f.k_d, f.s_d, f.dl_d, f.p_d, f.in_d, f.up) * \
        valid_pairs(f.out_h, f.k_h, f.s_h, f.dl_h, f.p_h, f.in_h, f.up) * \
        valid_pairs(f.out_w, f.k_w, f.s_w, f.dl_w, f.p_w, f.in_w, f.up) * f.cin * f.cout
    T = g.k_d * g.k_h * g.k_w
    xin = torch.randn((g.n, g.in_d, g.in_h, g.in_w, g.cin) if g.nd == 3 else (g.n, g.in_h, g.in_w, g.cin), device="cuda")
    yout = torch.randn((g.n, g.out_d, g.out_h, g.out_w, g.cout) if g.nd == 3 else (g.n, g.out_h, g.out_w, g.cout), device="cuda")
    if g.cin > 4:
        xin = xin.to(ops.ACT_DTYPE)
    if g.cout > 4:
        yout = yout.to(ops.ACT_DTYPE)
    wshape = ((g.k_d, g.k_h, g.k_w) if g.nd == 3 else (g.k_h, g.k_w)) + (g.cin, g.cout)
    w = torch.randn(wshape, device="cuda")
    bias = torch.randn(g.cout, device="cuda")
    if kind == "fwd":
        fn = lambda: ops.conv_fwd(xin, w, bias, g, 1, 0.3)
    elif kind == "dgrad":
        fn = lambda: ops.conv_dgrad(yout, w, g)
    else:
        fn = lambda: ops.conv_wgrad(xin, yout, g, wshape)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    M = f.n * f.out_d * f.out_h * f.out_w
    rows.append((us * cnt, kind, cnt, us, flops / us / 1e6, M, T * f.cin if kind != "wgrad" else T * g.cin, f.cout,
                 "nd%d in%dx%dx%d k%d s%d dl%d up%d" % (g.nd, g.in_d, g.in_h, g.in_w, g.k_h, g.s_h, g.dl_h, g.up)))
    del xin, yout, w
tot = sum(r[0] for r in rows)
print("total conv time per iteration (isolated): %.2f ms, %d distinct shapes, %d launches" % (tot / 1e3, len(rows), sum(r[2] for r in rows)))
print("%-6s %4s %9s %8s %9s %6s %5s  %-34s %6s" % ("kind", "cnt", "us/call", "TFLOP/s", "M", "K", "N", "geometry", "share"))
for r in sorted(rows, reverse=True):
    print("%-6s %4d %9.1f %8.1f %9d %6d %5d  %-34s %5.1f%%" % (r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], 100 * r[0] / tot))
