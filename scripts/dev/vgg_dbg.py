import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd.perceptual_loss import PerceptualLoss
from confignet_amd import ops
rng = np.random.default_rng(19)
pl = PerceptualLoss((64, 64, 3), "imagenet")
gt = torch.tensor(rng.uniform(-1, 1, size=(5, 64, 64, 3)), device="cuda", dtype=torch.float32)
gen = rng.uniform(-1, 1, size=(5, 64, 64, 3))
for wino in (True, False):
    ops.WINOGRAD = wino
    res = {}
    for rep in range(2):
        for fused in (False, True):
            pl.fused_tape = fused
            a = torch.tensor(gen, device="cuda", dtype=torch.float32, requires_grad=True)
            l1 = pl.loss(gt, a)
            (g1,) = torch.autograd.grad(l1 * 3.0, a)
            res[(fused, rep)] = g1.detach().clone()
    ref = res[(False, 0)]
    print("winograd", wino, "layerwise run-to-run %.3e  fused run-to-run %.3e  fused vs layerwise %.3e" % (
        float((res[(False, 1)] - ref).norm() / ref.norm()), float((res[(True, 1)] - res[(True, 0)]).norm() / ref.norm()),
        float((res[(True, 0)] - ref).norm() / ref.norm())))
    d = (res[(True, 0)] - ref).abs()
    print("  per-sample max abs diff", d.reshape(5, -1).max(1)[0].tolist(), " ref max", float(ref.abs().max()))
