"""How far is the implicit-GEMM tile from what the vendor sgemm reaches on the same M x K x N?  (torch.mm = rocBLAS/hipBLASLt fp32;
the 1x1 convolution below is the same product through igemm_fwd_kernel.)  Measurement only -- torch.mm is not on any product path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1000 / n


for m, k, n in [(65536, 3456, 192), (32768, 2304, 256), (4096, 2304, 256), (327680, 432, 96), (65536, 16384, 128), (16384, 1152, 128),
                (8192, 4608, 512), (4096, 1024, 256), (65536, 64, 256), (8192, 8192, 8192)]:
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda")
    fl = 2.0 * m * k * n
    us_v = t(lambda: torch.mm(a, b))
    h = 256 if m % 256 == 0 else 64
    g = ops.ConvSpec((1, 1)).geom((1, m // h, h, k), n)
    x4 = a.view(1, m // h, h, k); w4 = b.view(1, 1, k, n)
    us_c = t(lambda: ops.conv_fwd(x4, w4, None, g, 0, 0.0))
    print("M %7d K %6d N %5d: vendor sgemm %7.1f us %6.1f TF | igemm 1x1 %7.1f us %6.1f TF" % (m, k, n, us_v, fl / us_v / 1e6, us_c, fl / us_c / 1e6))
