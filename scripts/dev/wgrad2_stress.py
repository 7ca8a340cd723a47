"""The class-filter gradient of the generator's Conv3D 8^3 -> 16^3 layer (wgrad2_kernel) next to other kernels on a second stream:
any NaN / any difference from the first result?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
torch.manual_seed(0)
dev = "cuda"
r = lambda *s: torch.randn(*s, device=dev) * 0.05
cases = []
for (n, d, cin, cout) in ((8, 8, 256, 128), (8, 4, 512, 256)):
    g = ops.ConvSpec((3, 3, 3), up=1).geom((n, d, d, d, cin), cout)
    w = r(3, 3, 3, cin, cout)
    _, wd, _, g2 = ops.upfold_prepare(w, g)
    x, gy = r(n, d, d, d, cin), r(*ops.geom_out_shape(g))
    cases.append((g2, gy, x, tuple(wd.shape)))
g_a = ops.ConvSpec((3, 3)).geom((8, 64, 64, 64), 256)
xa, wa = r(8, 64, 64, 64), r(3, 3, 64, 256)
g_b = ops.ConvSpec((3, 3), stride=2).geom((8, 64, 64, 96), 192)
gyb, wb = r(*ops.geom_out_shape(g_b)), r(3, 3, 96, 192)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ops.WINOGRAD = False
for ci, (g2, a, b, shp) in enumerate(cases):
    ref = ops.conv_wgrad(a, b, g2, shp).clone()
    torch.cuda.synchronize()
    print("workspace bytes", int(ops.lib.cn_conv_wgrad_workspace_bytes(__import__("ctypes").byref(g2))), "Ktot x cout", shp);    print("case", ci, "reference finite", bool(torch.isfinite(ref).all()), "max", float(ref.abs().max()))
    for mode in ("with forward convs",):
        nan = diff = 0
        for rep in range(400):
            with torch.cuda.stream(sb):
                if mode == "with forward convs":
                    for _ in range(3): ops.conv_fwd(xa, wa, None, g_a)
                elif mode == "with parity dgrads":
                    for _ in range(3): ops.conv_dgrad(gyb, wb, g_b)
                elif mode == "two wgrads":
                    o2 = [ops.conv_wgrad(a, b, g2, shp) for _ in range(2)]
            with torch.cuda.stream(sa):
                outs = [ops.conv_wgrad(a, b, g2, shp) for _ in range(2)]
            torch.cuda.synchronize()
            for o in outs + (o2 if mode == "two wgrads" else []):
                nan += int(not bool(torch.isfinite(o).all()))
                bad = (o - ref).abs() > 1e-3 * float(ref.abs().max())
                if bool(bad.any()):
                    diff += 1
                    b2 = bad.reshape(-1, bad.shape[-1])
                    rows = torch.nonzero(b2.any(dim=1)).reshape(-1)
                    cols = torch.nonzero(b2.any(dim=0)).reshape(-1)
                    d = (o - ref).reshape(-1, bad.shape[-1])
                    print("    rep", rep, "bad entries", int(bad.sum()), "rows", int(rows.min()), "..", int(rows.max()), "(%d)" % rows.numel(),
                          "cols", int(cols.min()), "..", int(cols.max()), "(%d)" % cols.numel(),
                          "max |diff|", float(d.abs().max()), "o there", float(o.reshape(-1, bad.shape[-1])[rows[0], cols[0]]),
                          "ref there", float(ref.reshape(-1, bad.shape[-1])[rows[0], cols[0]]), flush=True)
        print("  %-22s launches with NaN %d, differing %d" % (mode, nan, diff), flush=True)
