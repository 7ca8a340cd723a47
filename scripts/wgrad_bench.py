"""Filter-gradient shapes of one second-stage iteration (256x256, batch 16), each timed in isolation: the round-3 kernel
(cn_conv_wgrad: split over rows, fp32 atomics) against cn_conv_wgrad_ws (wgrad2.hip: LDS-DMA main loop, partial slabs + ordered
reduction), results compared, and -- with `sweep` -- every tile x workgroup target of the new kernel (cn_conv_tune).
    python scripts/wgrad_bench.py [batch] [sweep] [geoms=FILE] [dump=FILE]
geoms=FILE: take the (geometry, count) list from FILE (written by an earlier run with dump=FILE) instead of tracing an iteration."""
import ctypes
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, ".")
from confignet_amd import ConfigNet, SyntheticFaceDataset, ops, optim
from confignet_amd._lib import lib
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs

import json
import os

FIELDS = [f[0] for f in ops.CnConvGeom._fields_]
calls = OrderedDict()
orig = ops.conv_wgrad
GEOMS = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("geoms=")), None)
DUMP = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("dump=")), None)
SWEEPOUT = next((open(a.split("=", 1)[1], "w") for a in sys.argv if a.startswith("sweepout=")), None)     # every sweep point: M K N tile target us


def conv_wgrad(x, gy, g, ws, out=None, **kw):
    if x.dtype == torch.float32 and gy.dtype == torch.float32:
        k = tuple(getattr(g, f) for f in FIELDS)
        calls[k] = calls.get(k, 0) + 1
    return orig(x, gy, g, ws, out=out, **kw)


B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
SWEEP = "sweep" in sys.argv
if GEOMS and os.path.exists(GEOMS):
    for k, cnt in json.load(open(GEOMS)):
        calls[tuple(k)] = cnt
else:
    ops.conv_wgrad = conv_wgrad
    np.random.seed(0)
    ds = SyntheticFaceDataset(64, 256, seed=1)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": B, "output_shape": (256, 256, 3)})
    ds.process_metadata(cfg, True)
    m = ConfigNet(cfg, seed=0)
    m.setup_training(None, ds, 0, real_training_set=ds)
    m.training_iteration(ds, ds, optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"]))
    torch.cuda.synchronize()
    ops.conv_wgrad = orig
    del m
    torch.cuda.empty_cache()
if DUMP:
    json.dump([[list(k), cnt] for k, cnt in calls.items()], open(DUMP, "w"))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


NS = next((int(a.split("=", 1)[1]) for a in sys.argv if a.startswith("ns=")), 0)
if NS:
    ops.check(lib.cn_conv_loop_select(-1, 0, NS, -1), "loop_select")      # stage count of the LDS-DMA loops
rows = []
tot_old = tot_new = tot_best = tot_prod = 0.0
for k, cnt in calls.items():
    g = ops.CnConvGeom(*k)
    taps = g.k_d * g.k_h * g.k_w
    Ktot = taps * g.cin
    M = g.n * g.out_d * g.out_h * g.out_w
    nbytes = int(lib.cn_conv_wgrad_workspace_bytes(ctypes.byref(g)))
    x = torch.randn((g.n, g.in_d, g.in_h, g.in_w, g.cin), device="cuda")
    gy = torch.randn((g.n, g.out_d, g.out_h, g.out_w, g.cout), device="cuda")
    gw_old = torch.zeros((Ktot, g.cout), device="cuda")
    gw_new = torch.zeros((Ktot, g.cout), device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def run_old(acc=1):
        ops.check(lib.cn_conv_wgrad(ctypes.byref(g), x.data_ptr(), gy.data_ptr(), gw_old.data_ptr(), acc, s), "old")

    def run_new(acc=1):
        nb = int(lib.cn_conv_wgrad_workspace_bytes(ctypes.byref(g)))
        ws = torch.empty(max(nb // 4, 1), device="cuda")
        ops.check(lib.cn_conv_wgrad_ws(ctypes.byref(g), x.data_ptr(), gy.data_ptr(), gw_new.data_ptr(), acc, ws.data_ptr(), nb, s), "new")

    run_old(0)
    run_new(0)
    torch.cuda.synchronize()
    err = float((gw_old - gw_new).abs().max() / (gw_old.abs().max() + 1e-30))
    t_old, t_new = timed(run_old), timed(run_new)
    # what the product launches for this geometry: ops.conv_wgrad (first-layer K = 27 and thin-output shapes have kernels of their own)
    w_shape = ((g.k_d,) if g.nd == 3 else ()) + (g.k_h, g.k_w, g.cin, g.cout)
    gw_prod = torch.zeros(w_shape, device="cuda")
    xp, gyp = (x, gy) if g.nd == 3 else (x[:, 0], gy[:, 0])
    t_prod = timed(lambda: ops.conv_wgrad(xp, gyp, g, w_shape, out=gw_prod))
    tot_prod += cnt * t_prod
    best = (t_new, "default")
    if SWEEP and nbytes >= 0:
        for tile in (0, 4, 2, 3, 5):                     # 128x128, 128x96, 64x64, 128x32, 256x64
            if tile == 3 and g.cout > 32:
                continue
            if tile == 5 and (g.cout % 64 or Ktot < 128):
                continue
            if tile == 4 and g.cout % 96:
                continue
            if tile == 0 and g.cout < 64:
                continue
            for target in (128, 256, 384, 512, 768, 1024, 1536, 2048):
                ops.check(lib.cn_conv_tune(tile, 0, target), "tune")
                try:
                    t = timed(run_new, 10)
                except Exception as e:
                    t = float("inf")
                if SWEEPOUT:
                    SWEEPOUT.write("%d %d %d %d %d %.1f\n" % (M, Ktot, g.cout, tile, target, t))
                if t < best[0]:
                    best = (t, "tile%d/wg%d" % (tile, target))
        ops.check(lib.cn_conv_tune(-1, 0, 0), "tune")
    flop = 2.0 * M * Ktot * g.cout
    rows.append((cnt * t_old, cnt, t_old, t_new, best, flop, M, Ktot, g.cout, nbytes, err, k, t_prod))
    tot_old += cnt * t_old
    tot_new += cnt * t_new
    tot_best += cnt * best[0]
    del x, gy, gw_old, gw_new
rows.sort(reverse=True)
print("filter gradients per iteration: round-3 kernel (cn_conv_wgrad) %.2f ms, cn_conv_wgrad_ws %.2f ms, PRODUCT dispatch (ops.conv_wgrad: + the K = 27 / thin-output kernels) %.2f ms%s"
      % (tot_old / 1e3, tot_new / 1e3, tot_prod / 1e3, ", best of sweep %.2f ms" % (tot_best / 1e3) if SWEEP else ""))
print("%3s %8s %8s %8s %7s %7s %8s %6s %5s %8s %9s  %s" % ("cnt", "old us", "new us", "prod us", "old TF", "new TF", "M", "K", "N", "ws MB", "rel err", "best"))
for _, cnt, t_old, t_new, best, flop, M, K, N, nb, err, k, t_prod in rows:
    print("%3d %8.1f %8.1f %8.1f %7.1f %7.1f %8d %6d %5d %8.1f %9.1e  %s %.1f  %s" % (cnt, t_old, t_new, t_prod, flop / t_old / 1e6, flop / t_new / 1e6, M, K, N, nb / 1e6, err, best[1], best[0], dict(zip(FIELDS, k)) if "geom" in sys.argv else ""))
