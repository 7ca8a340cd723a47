"""Fixed cost of a convolution launch: 1x1 layers with a tiny reduction depth (the product is negligible; what remains is the
launch, the gather prologue and the epilogue) next to the lab GEMM of the same size and to a plain elementwise pass over the output."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_lab", "gemm_lab.so"))
lab.gemm_lab.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]


def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1000 / n


s = torch.cuda.current_stream().cuda_stream
for m, n in [(4096, 256), (16384, 128), (65536, 256), (1024, 512)]:
    for k in (32, 64, 256, 1024, 2304):
        a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda"); c = torch.empty(m, n, device="cuda")
        g = ops.ConvSpec((1, 1)).geom((1, m // 64, 64, k), n)
        x4 = a.view(1, m // 64, 64, k); w4 = b.view(1, 1, k, n)
        us_c = t(lambda: ops.conv_fwd(x4, w4, None, g, 0, 0.0))
        us_l = t(lambda: lab.gemm_lab(6, a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, s))
        us_v = t(lambda: torch.mm(a, b, out=c))
        print("M %6d N %4d K %5d: igemm 1x1 %6.1f us | lab 64x64 %6.1f us | vendor %6.1f us" % (m, n, k, us_c, us_l, us_v))
