import ctypes, os, sys, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_lab.so"))
lib.gemm_lab.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
cfg, m, k, n = map(int, sys.argv[1:5])
a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda"); c = torch.empty(m, n, device="cuda")
for _ in range(10): lib.gemm_lab(cfg, a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
