import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
from confignet_amd.ops import ConvSpec
rng = np.random.default_rng(3)
C3 = ConvSpec((3, 3))
for (n, h, cin, cout) in [(5, 64, 64, 64), (5, 64, 3, 64), (5, 32, 64, 128), (5, 32, 128, 128), (5, 16, 128, 256), (5, 16, 256, 256), (5, 8, 256, 512), (5, 8, 512, 512),
                          (8, 32, 512, 512), (8, 64, 256, 256), (2, 64, 64, 64), (16, 64, 64, 64)]:
    x = torch.tensor(rng.normal(size=(n, h, h, cin)), device="cuda", dtype=torch.float32)
    w = torch.tensor(rng.normal(size=(3, 3, cin, cout)) * 0.05, device="cuda", dtype=torch.float32)
    b = torch.tensor(rng.normal(size=cout), device="cuda", dtype=torch.float32)
    g = C3.geom(tuple(x.shape), cout)
    gy = torch.tensor(rng.normal(size=(n, h, h, cout)), device="cuda", dtype=torch.float32)
    ops.WINOGRAD = False
    ref = ops.conv_fwd(x, w, b, g, ops.ACT_RELU)
    refd = ops.conv_dgrad(gy, w, g) if cin > 4 else None
    ops.WINOGRAD = True
    which = "F4" if ops._wino4_ok(g, cin, cout) else "F2" if ops._wino_ok(g, cin, cout) else "direct"
    outs = [ops.conv_fwd(x, w, b, g, ops.ACT_RELU) for _ in range(6)]
    torch.cuda.synchronize()
    rr = max(float((o - outs[0]).abs().max()) for o in outs)
    err = float((outs[0] - ref).abs().max() / ref.abs().max())
    msg = "fwd %-6s n=%d %dx%d %d->%d  run-to-run max %.3e  vs direct %.3e" % (which, n, h, h, cin, cout, rr, err)
    if refd is not None:
        whichd = "F4" if ops._wino4_ok(g, cout, cin) else "F2" if ops._wino_ok(g, cout, cin) else "direct"
        outs = [ops.conv_dgrad(gy, w, g) for _ in range(6)]
        torch.cuda.synchronize()
        rr = max(float((o - outs[0]).abs().max()) for o in outs)
        err = float((outs[0] - refd).abs().max() / refd.abs().max())
        msg += " | dgrad %-6s run-to-run %.3e vs direct %.3e" % (whichd, rr, err)
    print(msg)
