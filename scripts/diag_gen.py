import sys, numpy as np, torch
sys.path.insert(0, '.')
from confignet_amd.dnn_models.hologan_generator import HologanGenerator
from oracle import ref_nets as R
res, n = 128, 2
rng = np.random.default_rng(res)
g = HologanGenerator(43, (res, res), 128, 2, "tanh", rng=rng)
ws = g.get_weights()
r2 = np.random.default_rng(1)
for i, w in enumerate(ws):
    if w.ndim == 1: ws[i] = (w + r2.normal(size=w.shape) * 0.1).astype(np.float32)
g.set_weights(ws)
z = rng.normal(size=(n, 43)); rot = rng.uniform(-0.5, 0.5, size=(n, 3)).astype(np.float32); rot[:, 2] = 0
cot = rng.normal(size=(n, res, res, 3))
img = g((z, rot))
g.zero_grad()
torch.autograd.backward((img * torch.tensor(cot, device="cuda", dtype=torch.float32)).sum(), inputs=g.trainable_weights)
for dt in (torch.float64, torch.float32):
    wr = [torch.tensor(w, dtype=dt, requires_grad=True) for w in g.get_weights()]
    ref = R.generator_forward(wr, torch.tensor(z, dtype=dt), torch.tensor(rot, dtype=dt), res)
    grads = torch.autograd.grad((ref * torch.tensor(cot, dtype=dt)).sum(), wr, allow_unused=True)
    if dt == torch.float64: g64 = grads; print("img err vs f64", float((img.detach().cpu().double()-ref).abs().max()))
    else:
        print("CPU fp32 vs f64:")
        for i,(a,b) in enumerate(zip(grads, g64)):
            if a is None: continue
            print(i, tuple(a.shape), "%.2e" % float((a.double()-b).abs().max()/(b.abs().max()+1e-30)))
print("HIP vs f64:")
for i,(p,b) in enumerate(zip(g.weights, g64)):
    if b is None: continue
    print(i, g._entries[i][0], tuple(p.shape), "%.2e" % float((p.grad.cpu().double()-b).abs().max()/(b.abs().max()+1e-30)))
