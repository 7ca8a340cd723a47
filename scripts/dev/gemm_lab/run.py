"""Dev lab driver: correctness and TFLOP/s of the gemm_lab variants against torch.mm (vendor sgemm) on conv-sized products."""
import ctypes, os, sys, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_lab.so"))
lib.gemm_lab.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1000 / n


cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(12))
shapes = [(32768, 2304, 256), (8192, 8192, 8192), (65536, 3456, 256), (4096, 2304, 256), (16384, 1152, 128), (327680, 432, 128), (4096, 1024, 256)]
for m, k, n in shapes:
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda"); c = torch.empty(m, n, device="cuda")
    ref = torch.mm(a, b)
    fl = 2.0 * m * k * n
    line = "M %6d K %5d N %5d: vendor %6.1f TF |" % (m, k, n, fl / t(lambda: torch.mm(a, b)) / 1e6)
    s = torch.cuda.current_stream().cuda_stream
    for cfg in cfgs:
        c.zero_()
        rc = lib.gemm_lab(cfg, a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, s)
        if rc != 0:
            line += " c%d: n/a |" % cfg
            continue
        torch.cuda.synchronize()
        err = ((c - ref).abs().max() / ref.abs().max()).item()
        us = t(lambda: lib.gemm_lab(cfg, a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, s))
        line += " c%d: %5.1f%s |" % (cfg, fl / us / 1e6, "" if err < 1e-4 else " ERR %.1e" % err)
    print(line, flush=True)
