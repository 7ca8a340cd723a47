import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from confignet_amd import ops
rng = np.random.default_rng(0)
for shape in [(2, 16, 16, 64), (4, 64, 64, 32), (16, 128, 128, 64), (2, 8, 8, 8, 128)]:
    x = torch.tensor(rng.normal(size=shape), device="cuda", dtype=torch.float32)
    gy = torch.tensor(rng.normal(size=shape), device="cuda", dtype=torch.float32)
    y = ops.act_fwd(x, 1, 0.3)
    ref_gx = gy * torch.where(x > 0, 1.0, 0.3)
    ref_gb = ref_gx.reshape(-1, shape[-1]).sum(0)
    for dt in (torch.float32, torch.bfloat16):
        gx, gb = ops.act_bwd_bias(gy.to(dt), y.to(dt), 1, 0.3)
        r = ref_gx if dt == torch.float32 else (gy.to(dt).float() * torch.where(y.to(dt).float() > 0, 1.0, 0.3))
        print(shape, dt, gx.dtype, float((gx.float() - r).abs().max()), float((gb - r.reshape(-1, shape[-1]).sum(0)).abs().max() / ref_gb.abs().max()))
