"""Every dense-GEMM launch of one second-stage iteration (256x256, batch 16), timed per distinct shape."""
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, ".")
from confignet_amd import ConfigNet, SyntheticFaceDataset, ops, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs

calls = OrderedDict()
orig = ops.gemm


def gemm(a, b, trans_a=False, trans_b=False, bias=None, act=0, slope=0.0):
    k = (tuple(a.shape), tuple(b.shape), bool(trans_a), bool(trans_b), bias is not None, act)
    calls[k] = calls.get(k, 0) + 1
    return orig(a, b, trans_a, trans_b, bias, act, slope)


ops.gemm = gemm
np.random.seed(0)
ds = SyntheticFaceDataset(64, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 16, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
m.setup_training(None, ds, 0, real_training_set=ds)
m.training_iteration(ds, ds, optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"]))
torch.cuda.synchronize()
ops.gemm = orig
del m
torch.cuda.empty_cache()
rows = []
for k, cnt in calls.items():
    sa, sb, ta, tb, hb, act = k
    a, b = torch.randn(sa, device="cuda"), torch.randn(sb, device="cuda")
    mm = sa[1] if ta else sa[0]
    kk = sa[0] if ta else sa[1]
    nn = sb[0] if tb else sb[1]
    bias = torch.randn(nn, device="cuda") if hb else None
    fn = lambda: orig(a, b, ta, tb, bias, act, 0.2)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    mb = 4e-6 * (mm * kk + kk * nn + mm * nn)
    rows.append((us * cnt, cnt, us, 2e-6 * mm * nn * kk / us, mb / us * 1e3, mm, nn, kk, ta, tb))
tot = sum(r[0] for r in rows)
print("total dense GEMM time per iteration (isolated): %.2f ms, %d distinct, %d launches" % (tot / 1e3, len(rows), sum(r[1] for r in rows)))
print("%4s %9s %8s %8s %8s %8s %8s  ta tb" % ("cnt", "us/call", "TFLOP/s", "GB/s", "M", "N", "K"))
for r in sorted(rows, reverse=True)[:40]:
    print("%4d %9.1f %8.2f %8.0f %8d %8d %8d  %d  %d  (%.1f%%)" % (r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], 100 * r[0] / tot))
