"""Dev: compare the experimental bf16x3 convolution (CN_BF16X3=1) with the fp32-MFMA path on one shape: error and time."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from confignet_amd import ops
    n, h, w, cin, cout, k = map(int, sys.argv[2:8])
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(n, h, w, cin, device="cuda", generator=gen)
    wt = torch.randn(k, k, cin, cout, device="cuda", generator=gen) / np.sqrt(k * k * cin)
    b = torch.randn(cout, device="cuda", generator=gen)
    g = ops.ConvSpec((k, k)).geom(tuple(x.shape), cout)
    y = ops.conv_fwd(x, wt, b, g, 0, 0.0)
    for _ in range(3): ops.conv_fwd(x, wt, b, g, 0, 0.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv_fwd(x, wt, b, g, 0, 0.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    # float64 reference of a slice of rows through torch conv on the device
    xr = x[:1].permute(0, 3, 1, 2).double(); wr = wt.permute(3, 2, 0, 1).double()
    total = k - 1
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xr, (total // 2, total - total // 2, total // 2, total - total // 2)), wr, b.double())
    err = float((y[:1].permute(0, 3, 1, 2).double() - ref).abs().max())
    print("%s: %.1f us, %.1f TFLOP/s, max abs err vs float64 %.3e (output scale %.2f)" % (
        os.environ.get("CN_BF16X3", "0"), us, 2.0 * n * h * w * k * k * cin * cout / us / 1e6, err, float(ref.abs().max())))
    sys.exit(0)
for shape in ("16 64 64 256 256 3", "16 128 128 128 128 3", "16 32 32 512 512 3", "16 256 256 64 64 3"):
    for mode in ("0", "1", "3"):     # fp32 MFMA / 2-term split (3 MFMAs) / 3-term split (6 MFMAs)
        env = dict(os.environ, CN_BF16X3=mode)
        out = subprocess.run([sys.executable, __file__, "child"] + shape.split(), env=env, capture_output=True, text=True)
        print(shape, "| bf16x3 =", (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1])
