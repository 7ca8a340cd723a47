"""Tile-config / split-K sweep on representative implicit-GEMM shapes (tuning aid; uses the CN_CFG / CN_SPLITS
overrides of libconfignet_hip.so).  cfg: 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 128x32, 4 = 128x96.
`python scripts/conv_tune.py [fwd|dgrad]`."""
import os, sys, subprocess, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # (n, h, w, cin, cout, k, stride)
    (16, 64, 64, 96, 192, 3, 2), (16, 128, 128, 48, 96, 3, 2), (16, 32, 32, 192, 384, 3, 2), (16, 16, 16, 384, 768, 3, 2),
    (96, 16, 16, 384, 768, 3, 2), (8, 16, 16, 1024, 256, 1, 1), (8, 16, 16, 256, 1024, 1, 1), (8, 32, 32, 512, 128, 1, 1),
    (8, 32, 32, 128, 512, 1, 1), (8, 8, 8, 2048, 512, 1, 1), (8, 8, 8, 512, 2048, 1, 1), (8, 64, 64, 64, 256, 1, 1),
    (8, 64, 64, 256, 64, 1, 1), (8, 32, 32, 256, 512, 1, 2), (8, 16, 16, 256, 256, 3, 1), (8, 8, 8, 512, 512, 3, 1),
]
KIND = os.environ.get("CN_TUNE_KIND", "fwd")
if len(sys.argv) > 1 and sys.argv[1] in ("fwd", "dgrad"):
    KIND = os.environ["CN_TUNE_KIND"] = sys.argv[1]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from confignet_amd import ops
    res = {}
    for i, (n, h, w, cin, cout, k, st) in enumerate(SHAPES):
        spec = ops.ConvSpec((k, k), stride=st)
        x = torch.randn(n, h, w, cin, device="cuda"); wt = torch.randn(k, k, cin, cout, device="cuda"); b = torch.randn(cout, device="cuda")
        g = spec.geom(tuple(x.shape), cout)
        if KIND == "dgrad":
            gy = torch.randn(n, g.out_h, g.out_w, cout, device="cuda")
            fn = lambda: ops.conv_dgrad(gy, wt, g)
        else:
            fn = lambda: ops.conv_fwd(x, wt, b, g, 2, 0.0)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res[i] = e0.elapsed_time(e1) * 100
    print(json.dumps(res))
    sys.exit(0)
table = {}
for cfg in ("", "0", "1", "2", "4"):
    for sp in ("", "1", "2", "4", "8", "16"):
        env = dict(os.environ)
        if cfg: env["CN_CFG"] = cfg
        if sp: env["CN_SPLITS"] = sp
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        table[(cfg or "auto", sp or "auto")] = json.loads(out)
for i, sh in enumerate(SHAPES):
    n, h, w, cin, cout, k, st = sh
    M = n * (h // st) * (w // st); flops = 2.0 * M * k * k * cin * cout
    best = min(table.items(), key=lambda kv: kv[1][str(i)])
    auto = table[("auto", "auto")][str(i)]
    if KIND == "dgrad":
        M, cin, cout = n * h * w, cout, cin          # (flops are the same as the forward's)
    print("M=%-6d K=%-5d N=%-4d auto %7.1f us (%5.1f TF) | best cfg=%s splits=%s %7.1f us (%5.1f TF)" % (
        M, k * k * cin, cout, auto, flops / auto / 1e6, best[0][0], best[0][1], best[1][str(i)], flops / best[1][str(i)] / 1e6))
    print("     " + "  ".join("%s/%s:%.0f" % (c, s_, table[(c, s_)][str(i)]) for c in ("0", "1", "2", "4") for s_ in ("1", "2", "4", "8", "16")))
