import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
ops.set_activation_dtype("bf16")
cases = [((2, 16, 16, 64), (4, 4), 32, 1, 1), ((2, 16, 16, 512), (4, 4), 256, 1, 0), ((3, 16, 16, 1024), (1, 1), 512, 1, 0), ((16, 32, 32, 96), (3, 3), 192, 2, 0),
         ((4, 64, 64, 64), (3, 3), 64, 1, 0), ((16, 32, 32, 128), (3, 3), 256, 1, 0)]
out = {}
for ci, (xs, k, cout, stride, up) in enumerate(cases):
    rng = np.random.default_rng(ci)
    x = torch.tensor(rng.normal(size=xs).astype(np.float32)).to(torch.bfloat16).cuda()
    w = torch.tensor((rng.normal(size=(*k, xs[-1], cout)) / np.sqrt(np.prod(k) * xs[-1])).astype(np.float32)).cuda()
    b = torch.tensor(rng.normal(size=cout).astype(np.float32)).cuda()
    g = ops.ConvSpec(k, stride=stride, up=up).geom(xs, cout)
    y = ops.conv_fwd(x, w, b, g, 1, 0.3)
    gy = torch.tensor(rng.normal(size=tuple(y.shape)).astype(np.float32)).to(torch.bfloat16).cuda()
    gu = ops.conv_dgrad(gy, w, g)
    torch.cuda.synchronize()
    out["y%d" % ci] = y.float().cpu().numpy(); out["gu%d" % ci] = gu.float().cpu().numpy()
path = sys.argv[1]
if os.path.exists(path):
    ref = np.load(path)
    for k_, v in out.items():
        r = ref[k_]
        bad = ~np.isfinite(v)
        d = np.abs(np.nan_to_num(v) - r)
        rows = d.reshape(-1, d.shape[-1]).max(axis=1)
        print(k_, v.shape, "nan", int(bad.sum()), "max diff %.4f of scale %.3f" % (d.max(), np.abs(r).max()), "bad rows", int((rows > 0.05 * np.abs(r).max()).sum()), "of", rows.size,
              "first bad rows", np.nonzero(rows > 0.05 * np.abs(r).max())[0][:12])
else:
    np.savez(path, **out)
    print("saved")
