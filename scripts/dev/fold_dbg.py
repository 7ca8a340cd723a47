"""Debug: the folded-tape ResNet-50 backward against the composite form, piece by piece."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops
from confignet_amd.dnn_models.real_encoder import RealEncoder
rng = np.random.default_rng(11)
enc = RealEncoder(43, (64, 64, 3), ((-30, 30), (-10, 10), (0, 0)), rng=rng)
with torch.no_grad():
    enc.arena.add_(torch.tensor(rng.normal(size=enc.arena.shape) * 0.02, device="cuda", dtype=torch.float32))
    for w in enc.weights:
        if not w.requires_grad:
            w.copy_(torch.tensor(rng.uniform(0.5, 1.5, size=tuple(w.shape)), device="cuda", dtype=torch.float32))
enc.mark_updated(); enc.non_trainable_changed()
img_np = rng.uniform(-1, 1, size=(3, 64, 64, 3)).astype(np.float32)
cap = {}
orig = ops.bn_fold_bwd
def spy(seg9, blocks, gwf, gsh, arena, a, rs, bm, gout):
    cap.update(seg9=seg9.cpu().numpy(), gwf=gwf.clone(), gsh=gsh.clone(), a=a.clone(), rs=rs.clone(), bm=bm.clone(), before=gout.clone())
    orig(seg9, blocks, gwf, gsh, arena, a, rs, bm, gout)
    cap["after"] = gout.clone()
ops.bn_fold_bwd = spy
res = {}
for folded in (False, True):
    enc.folded_tape = folded
    img = torch.tensor(img_np, device="cuda")
    emb, rot = enc(img)
    loss = (emb ** 2).sum() + (rot ** 2).sum() * 10
    grads = torch.autograd.grad(loss, enc.trainable_weights)
    res[folded] = [g.clone() for g in grads]
torch.cuda.synchronize()
names = [enc._entries[i][0] for i in enc._trainable_idx]
bad = 0
for n, a, b in zip(names, res[False], res[True]):
    e = float((a - b).norm() / (a.norm() + 1e-30))
    if e > 1e-3:
        bad += 1
        if bad < 25:
            print("%-40s %-22s rel %.3e  |ref| %.3e |got| %.3e" % (n, tuple(a.shape), e, float(a.norm()), float(b.norm())))
print("bad tensors: %d of %d" % (bad, len(names)))
# the kernel against torch on the captured operands
seg = cap["seg9"]
base = 0
worst = 0.0
for r in seg:
    src, dst, numel, cout, aoff, boff, goff, beoff, blk = map(int, r)
    K = numel // cout
    g = cap["gwf"][dst:dst + numel].view(K, cout).double()
    w = enc.arena[src:src + numel].view(K, cout).double()
    a_ = cap["a"][aoff:aoff + cout].double(); rs_ = cap["rs"][aoff:aoff + cout].double(); bm_ = cap["bm"][aoff:aoff + cout].double()
    gs = cap["gsh"][aoff:aoff + cout].double()
    d = (cap["after"] - cap["before"]).double()
    for what, got, want in (("kernel", d[src:src + numel].view(K, cout), g * a_), ("gamma", d[goff:goff + cout], rs_ * ((g * w).sum(0) + gs * bm_)),
                            ("beta", d[beoff:beoff + cout], gs), ("bias", d[boff:boff + cout], a_ * gs)):
        e = float((got - want).norm() / (want.norm() + 1e-30))
        worst = max(worst, e)
        if e > 1e-4:
            print("fold kernel: segment at %d %s rel %.3e" % (src, what, e))
print("fold kernel vs torch: worst %.3e; gwf finite %s gsh finite %s" % (worst, bool(torch.isfinite(cap["gwf"]).all()), bool(torch.isfinite(cap["gsh"]).all())))
