import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import functional as F
rng = np.random.default_rng(21)
shapes = [(16, 1), (16, 1), (7, 1), (16, 1), (33, 1), (16, 1)]
labels = [1.0, 0.0, 1.0, 1.0, 0.0, 0.0]
weights = [1.0, 0.5, -2.0, 0.0, 3.0, 1.0]
scs = [rng.normal(size=s) * 3 for s in shapes]
dev = lambda a: torch.tensor(np.asarray(a), device="cuda", dtype=torch.float32)
a = [dev(s).requires_grad_(True) for s in scs]
b = [dev(s).requires_grad_(True) for s in scs]
la = F.gan_losses(a, labels)
lb = [F.gan_loss(x, l) for x, l in zip(b, labels)]
used = [0, 1, 2, 4, 5]
ga = torch.autograd.grad(sum(weights[j] * la[j] for j in used), [a[j] for j in used])
gb = torch.autograd.grad(sum(weights[j] * lb[j] for j in used), [b[j] for j in used])
for j, x, y in zip(used, ga, gb):
    print(j, (x / y).reshape(-1)[:5].tolist())
