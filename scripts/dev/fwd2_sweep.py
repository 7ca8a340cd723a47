"""Sweep of the forward / data-gradient main loops over the iteration's own shapes: for every captured conv_fwd / conv_dgrad
launch with >= 64 channels on both sides that Winograd does not take, time tile cfg x split-K for the register-staged loop
(gemm1x1.hip) and the LDS-DMA loop (fwd2.hip, NS = 3 / 4).  python scripts/dev/fwd2_sweep.py [batch] > out.txt"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import os as _os
NP = int(_os.environ.get("SWEEP_NP", "-1"))
KBS = int(_os.environ.get("SWEEP_KB", "0"))

sys.path.insert(0, ".")
from confignet_amd import ConfigNet, SyntheticFaceDataset, ops, optim
from confignet_amd._lib import lib
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs

FIELDS = [f[0] for f in ops.CnConvGeom._fields_]
calls = OrderedDict()
orig = {k: getattr(ops, k) for k in ("conv_fwd", "conv_dgrad")}


def rec(kind, g):
    k = (kind,) + tuple(getattr(g, f) for f in FIELDS)
    calls[k] = calls.get(k, 0) + 1


def conv_fwd(x, w, bias, g, act=0, slope=0.0):
    rec("fwd", g)
    return orig["conv_fwd"](x, w, bias, g, act, slope)


def conv_dgrad(gy, wt, g):
    rec("dgrad", g)
    return orig["conv_dgrad"](gy, wt, g)


ops.conv_fwd, ops.conv_dgrad = conv_fwd, conv_dgrad
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
np.random.seed(0)
ds = SyntheticFaceDataset(64, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": B, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
m.setup_training(None, ds, 0, real_training_set=ds)
m.training_iteration(ds, ds, optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"]))
torch.cuda.synchronize()
ops.conv_fwd, ops.conv_dgrad = orig["conv_fwd"], orig["conv_dgrad"]
del m
torch.cuda.empty_cache()


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


CFGS = (-1, 0, 1, 2, 4)
SPLITS = (0, 1, 2, 4, 8)
tot = {"old_auto": 0.0, "new_auto3": 0.0, "new_auto4": 0.0, "old_best": 0.0, "new_best": 0.0, "either_best": 0.0}
for kk, cnt in calls.items():
    kind = kk[0]
    g = ops.CnConvGeom(*kk[1:])
    MINC = int(os.environ.get('SWEEP_MINC', '64'))
    MAXC = int(os.environ.get('SWEEP_MAXC', '100000'))
    if min(g.cin, g.cout) < MINC or min(g.cin, g.cout) > MAXC or g.cin % 16 or g.cout % 16:
        continue
    if ops._wino_ok(g, g.cin, g.cout) if kind == "fwd" else ops._wino_ok(g, g.cout, g.cin):
        continue                                     # Winograd's
    xin = torch.randn((g.n, g.in_d, g.in_h, g.in_w, g.cin) if g.nd == 3 else (g.n, g.in_h, g.in_w, g.cin), device="cuda")
    yout = torch.randn((g.n, g.out_d, g.out_h, g.out_w, g.cout) if g.nd == 3 else (g.n, g.out_h, g.out_w, g.cout), device="cuda")
    wshape = ((g.k_d, g.k_h, g.k_w) if g.nd == 3 else (g.k_h, g.k_w)) + (g.cin, g.cout)
    w = torch.randn(wshape, device="cuda")
    bias = torch.randn(g.cout, device="cuda")
    fn = (lambda: ops.conv_fwd(xin, w, bias, g, 1, 0.3)) if kind == "fwd" else (lambda: ops.conv_dgrad(yout, w, g))
    res = {}
    for loop, ns in ((0, 0), (1, 3), (1, 4)):
        ops.check(lib.cn_conv_loop_select(loop, KBS, ns, NP), "select")
        for c in CFGS:
            for s in SPLITS:
                ops.check(lib.cn_conv_tune(c, s, 0), "tune")
                try:
                    res[(loop, ns, c, s)] = timed(fn)
                except Exception as e:      # unsupported combination
                    res[(loop, ns, c, s)] = float("inf")
    ops.check(lib.cn_conv_tune(-1, 0, 0), "tune")
    ops.check(lib.cn_conv_loop_select(-1, 0, 0, -1), "select")
    old = {k: v for k, v in res.items() if k[0] == 0}
    new = {k: v for k, v in res.items() if k[0] == 1}
    bo, bn = min(old, key=old.get), min(new, key=new.get)
    M = g.n * g.out_d * g.out_h * g.out_w if kind == "fwd" else g.n * (g.in_d << g.up if g.nd == 3 else 1) * (g.in_h << g.up) * (g.in_w << g.up)
    T = g.k_d * g.k_h * g.k_w
    K, N = (T * g.cin, g.cout) if kind == "fwd" else (T * g.cout, g.cin)
    a0, a3, a4 = res[(0, 0, -1, 0)], res[(1, 3, -1, 0)], res[(1, 4, -1, 0)]
    print("%-5s x%-2d M %7d K %5d N %4d nd%d k%d s%d up%d | old auto %6.1f  new auto ns3 %6.1f ns4 %6.1f | old best cfg %2d sp %d: %6.1f | "
          "new best ns%d cfg %2d sp %d: %6.1f" % (kind, cnt, M, K, N, g.nd, g.k_h, g.s_h, g.up, a0, a3, a4, bo[2], bo[3], old[bo], bn[1], bn[2], bn[3], new[bn]))
    print("      new ns3: " + "  ".join("c%d/s%d:%.0f" % (c, s, res[(1, 3, c, s)]) for c in CFGS[1:] for s in SPLITS[1:] if res[(1, 3, c, s)] < 1e9))
    sys.stdout.flush()
    tot["old_auto"] += cnt * a0; tot["new_auto3"] += cnt * a3; tot["new_auto4"] += cnt * a4
    tot["old_best"] += cnt * old[bo]; tot["new_best"] += cnt * new[bn]; tot["either_best"] += cnt * min(old[bo], new[bn])
    del xin, yout, w
print({k: round(v / 1e3, 3) for k, v in tot.items()}, "ms per iteration")
