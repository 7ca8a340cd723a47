"""Sweep tile configuration x split-K (fwd / dgrad) and the workgroup target (wgrad) over every distinct implicit-GEMM launch
of one second-stage iteration (256x256, batch 16, fp32): what the heuristic in cn_conv_fwd leaves on the table.
usage: python scripts/conv_sweep.py [min_share_percent]"""
import sys
import torch
sys.path.insert(0, ".")
min_share = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4
sys.argv = sys.argv[:1]
import scripts.conv_shapes_bench as B          # runs the capture + the baseline table (prints it)
from confignet_amd import ops
from confignet_amd._lib import lib



def time_fn(fn, reps=5):
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


tot = sum(r[0] for r in B.rows)
gain = 0.0
print("\nsweep (shapes with >= %.1f%% of the conv time)" % min_share)
for kk, cnt in B.calls.items():
    kind = kk[0]
    g = ops.CnConvGeom(*kk[1:])
    if g.cout <= 4 or g.cin <= 4:
        continue
    xin = torch.randn((g.n, g.in_d, g.in_h, g.in_w, g.cin) if g.nd == 3 else (g.n, g.in_h, g.in_w, g.cin), device="cuda")
    yout = torch.randn((g.n, g.out_d, g.out_h, g.out_w, g.cout) if g.nd == 3 else (g.n, g.out_h, g.out_w, g.cout), device="cuda")
    wshape = ((g.k_d, g.k_h, g.k_w) if g.nd == 3 else (g.k_h, g.k_w)) + (g.cin, g.cout)
    w = torch.randn(wshape, device="cuda")
    if kind == "fwd":
        fn = lambda: ops.conv_fwd(xin, w, None, g, 0, 0.0)
    elif kind == "dgrad":
        fn = lambda: ops.conv_dgrad(yout, w, g)
    else:
        fn = lambda: ops.conv_wgrad(xin, yout, g, wshape)
    lib.cn_conv_tune(-1, 0, 0)
    auto = time_fn(fn)
    if auto * cnt / tot * 100 < min_share:
        continue
    res = {}
    if kind == "wgrad":
        for wb in (512, 1024, 4096, 8192):
            lib.cn_conv_tune(-1, 0, wb)
            res["wg%d" % wb] = time_fn(fn)
    else:
        for cfg in (0, 1, 2, 4):
            if cfg == 4 and (g.cout if kind == "fwd" else g.cin) % 96:
                continue
            for sp in (1, 2, 4, 8):
                lib.cn_conv_tune(cfg, sp, 0)
                try:
                    res["c%ds%d" % (cfg, sp)] = time_fn(fn)
                except Exception:
                    pass
    lib.cn_conv_tune(-1, 0, 0)
    best = min(res.items(), key=lambda kv: kv[1])
    M = g.n * g.out_d * g.out_h * g.out_w
    gain += max(0.0, auto - best[1]) * cnt
    print("%-6s x%-3d M=%-7d cin=%-4d cout=%-4d k%d s%d dl%d nd%d  auto %7.1f us | best %-6s %7.1f us (%+.0f%%)  %s" % (
        kind, cnt, M, g.cin, g.cout, g.k_h, g.s_h, g.dl_h, g.nd, auto, best[0], best[1], 100 * (best[1] - auto) / auto,
        " ".join("%s:%.0f" % kv for kv in sorted(res.items(), key=lambda kv: kv[1])[:4])))
print("total possible gain over the heuristic: %.2f ms of %.2f ms" % (gain / 1e3, tot / 1e3))
