"""GPU parity of the networks, losses and training steps (HIP path through the C ABI) against the
CPU oracle in float64 on identical seeded weights and inputs.  Tolerance: outputs / loss scalars
1e-3 max-abs (north_star), gradients 2e-3 of the tensor's max magnitude."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_nets as R
from oracle import ref_ops as O
from oracle import ref_steps as S

FM = OrderedDict([("beard_style_embedding", (9, 7)), ("blendshape_values", (62, 30)), ("eye_color", (8, 3)),
                  ("head_hair_color", (3, 3))])          # latent_dim 43


def w64(net, grad=True):
    return [torch.tensor(w, dtype=torch.float64, requires_grad=grad) for w in net.get_weights()]


def t64(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def close(got, ref, tol=1e-3, what="", rel=False):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = ref.detach().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30) if rel else max(1.0, np.abs(ref).max())
    err = np.abs(got - ref).max()
    assert np.isfinite(got).all(), what + ": non-finite"
    assert err <= tol * scale, "%s: max abs err %.3e vs scale %.3e" % (what, err, scale)


def _grad_pairs(parts, ref_grads):
    """[(label, product gradient, oracle gradient)] over the trainable weights of parts = [(net, what, slice into ref_grads)]."""
    out = []
    for net, what, sl in parts:
        for i, (p, g) in enumerate(zip(net.weights, ref_grads[sl])):
            if not p.requires_grad:
                continue
            if g is None:
                g = torch.zeros(tuple(p.shape), dtype=torch.float64)
            out.append(("%s grad[%d] %s" % (what, i, tuple(p.shape)), p.grad.detach().cpu().double(), g.detach()))
    return out


def _rel_errors(pairs):
    errs = []
    for label, got, g in pairs:
        assert torch.isfinite(got).all(), label + ": non-finite"
        den = float(g.norm())
        if den == 0.0:
            assert float(got.abs().max()) == 0.0, label + " should be exactly zero"
            continue
        errs.append((float((got - g).norm()) / den, label))
    return errs


def ops_mod():
    from confignet_amd import ops
    return ops


def check_grads(parts, recompute, tol=5e-3, max_flips=12, more_flips=48):
    """Gradient parity at `tol` relative L2 per tensor, with branch decisions accounted for by name.

    A LeakyReLU / ReLU whose pre-activation is within fp32 rounding of zero takes the other branch on the GPU than in the
    float64 oracle.  The forward value is continuous there, but the derivative mask differs in that element, which moves
    every upstream gradient by ~1/sqrt(#elements) (3e-3..6e-2 measured, depending on depth) -- far above the kernels' own
    error (2e-4 in test_ops_gpu).  Instead of a tolerance that absorbs it, the oracle lists its near-zero pre-activations
    (oracle.ref_ops.BranchControl), each candidate is flipped in a separate oracle run, and the product gradient must equal
    the oracle gradient plus the single-flip differences of a subset of them (gradients are linear in each mask element),
    to `tol`.  recompute() reruns the oracle and returns its gradient list.  The `max_flips` nearest candidates are tried
    first; when they do not explain the difference (the accumulation order of the atomically added partial sums differs from
    run to run, so which near-zero elements flip does too) the fit is repeated once over the `more_flips` nearest."""
    O.BranchControl.start(record=True)
    try:
        base = recompute()
        # near-zero pre-activations that a gradient actually reaches, closest to zero first
        all_cands = sorted((c for c in O.BranchControl.candidates() if c[3] > 0.0), key=lambda t: t[2])
        cands = all_cands[:max_flips]
    finally:
        O.BranchControl.stop()
    pairs = _grad_pairs(parts, base)
    errs = _rel_errors(pairs)
    print("check_grads: worst rel-L2 before branch accounting %.3e (%s), tol %.1e" % (max(errs)[0], max(errs)[1], tol))
    if all(e <= tol for e, _ in errs):
        return
    assert cands, "gradient mismatch with no near-zero pre-activation to attribute it to: %s" % sorted(errs, reverse=True)[:3]
    # single-flip differences, every tensor scaled by its oracle norm so that each weighs the same in the fit
    live = [(label, got, g, float(g.norm())) for label, got, g in pairs if float(g.norm()) > 0.0]
    resid = torch.cat([((got - g) / den).reshape(-1) for _, got, g, den in live])
    cols = []

    def fit_flips(cands):
        for cid, idx, _, _ in cands[len(cols):]:
            O.BranchControl.start(flips=[(cid, idx)])
            try:
                flipped = _grad_pairs(parts, recompute())
            finally:
                O.BranchControl.stop()
            fl = {label: g for label, _, g in flipped}
            cols.append(torch.cat([((fl[label] - g) / den).reshape(-1) for label, _, g, den in live]))
        # a flip is taken or not: greedy selection of the single-flip differences (coefficient exactly 1) that reduce the residual
        fit, chosen = torch.zeros_like(resid), []
        while True:
            cur = float((resid - fit).norm())
            gains = [(cur - float((resid - fit - c).norm()), k) for k, c in enumerate(cols) if k not in chosen]
            if not gains or max(gains)[0] <= 1e-3 * cur:
                break
            k = max(gains)[1]
            chosen.append(k)
            fit = fit + cols[k]
        rels, o = [], 0
        for label, got, g, den in live:
            n = g.numel()
            rels.append((float((resid[o:o + n] - fit[o:o + n]).norm()), label))
            o += n
        print("check_grads: worst rel-L2 after %d of %d flips %.3e" % (len(chosen), len(cols), max(rels)[0]))
        return rels, chosen

    rels, chosen = fit_flips(cands)
    if max(rels)[0] > tol and len(all_cands) > len(cands):
        cands = all_cands[:more_flips]
        rels, chosen = fit_flips(cands)
    coef = [1 if k in chosen else 0 for k in range(len(cols))]
    for rel, label in rels:
        assert rel <= tol, "%s: rel-L2 %.3e with %d of the %d nearest branch decisions flipped %s" % (
            label, rel, len(chosen), len(cands), coef)


def check_grads_forced(parts, recompute, log, tol=5e-3):
    """Gradient parity with EVERY branch decision forced: `log` holds the decisions the product's backward pass took
    (confignet_amd.ops.branch_log: the sign masks its LeakyReLU / ReLU derivatives were taken from, the fp32 inputs of the
    max-pools it differentiated); the oracle pass takes them over (oracle.ref_ops.BranchControl(forced=...): a forced decision
    may differ from the oracle's own only on a near-zero input / near-tied window, and every activation or pool that a gradient
    reaches must find its logged counterpart).  What is left between the two gradients is fp32 summation error: the whole-step
    chains are held at the 5e-3 of the single networks, deterministically -- no greedy fit, no run-to-run spread."""
    O.BranchControl.start(forced=log)
    try:
        ref = recompute()
        rep = O.BranchControl.forced_report()
    finally:
        O.BranchControl.stop()
    print("check_grads_forced: %d calls forced, %d decisions differ from the oracle's own (largest margin %.2e), %d unmatched" %
          (rep["forced_calls"], rep["forced_decisions"], rep["max_margin"], len(rep["unmatched"])))
    assert not rep["unmatched"], "oracle activations / pools reached by a gradient without a logged product decision: %s" % rep["unmatched"][:4]
    assert rep["forced_calls"] > 0
    errs = _rel_errors(_grad_pairs(parts, ref))
    print("check_grads_forced: worst rel-L2 %.3e (%s), tol %.1e" % (max(errs)[0], max(errs)[1], tol))
    for e, label in errs:
        assert e <= tol, "%s: rel-L2 %.3e with the product's branch decisions forced" % (label, e)


def close_grads(net, ref_grads, what, tol=5e-3, recompute=None):
    """One network's gradients against the oracle's at `tol` relative L2 per tensor.  recompute (a callable returning the
    oracle gradient list again) enables the branch-flip accounting of check_grads()."""
    if recompute is not None:
        return check_grads([(net, what, slice(None))], recompute, tol)
    errs = _rel_errors(_grad_pairs([(net, what, slice(None))], list(ref_grads)))
    for e, label in errs:
        assert e <= tol, "%s: rel-L2 %.3e" % (label, e)


def randomize(net, seed, scale=0.1):
    """Biases / gammas / betas are zero/one-initialised: perturb them so their gradients matter."""
    rng = np.random.default_rng(seed)
    ws = net.get_weights()
    for i, w in enumerate(ws):
        if w.ndim == 1:
            ws[i] = (w + rng.normal(size=w.shape) * scale).astype(np.float32)
    ws_ok = ws
    net.set_weights(ws_ok)


@pytest.mark.parametrize("res,n", [(128, 2), (256, 1)])
def test_generator_forward_backward(res, n):
    from confignet_amd.dnn_models.hologan_generator import HologanGenerator
    rng = np.random.default_rng(res)
    g = HologanGenerator(43, (res, res), 128, 2, "tanh", rng=rng)
    randomize(g, 1)
    z = rng.normal(size=(n, 43))
    rot = rng.uniform(-0.5, 0.5, size=(n, 3)).astype(np.float32)
    rot[:, 2] = 0
    rot_t = torch.tensor(rot, device="cuda", requires_grad=True)
    img = g((z, rot_t))
    wr = w64(g)
    rot_r = t64(rot).requires_grad_(True)
    ref = R.generator_forward(wr, t64(z), rot_r, res)
    close(img, ref, what="generator image")
    cot = rng.normal(size=tuple(ref.shape))
    g.zero_grad()
    torch.autograd.backward((img * torch.tensor(cot, device="cuda", dtype=torch.float32)).sum(),
                            inputs=g.trainable_weights + [rot_t])
    grads = torch.autograd.grad((ref * t64(cot)).sum(), wr + [rot_r], allow_unused=True)
    close_grads(g, None, "generator", recompute=lambda: torch.autograd.grad(
        (R.generator_forward(wr, t64(z), rot_r, res) * t64(cot)).sum(), wr, allow_unused=True))
    close(rot_t.grad, grads[-1], tol=2e-2, what="d/d rotation", rel=True)
    # predict() == eager call, numpy out; learned_input kernel gradient is identically zero
    # (fp32 atomics in the statistics kernels: the last bits are run-dependent)
    np.testing.assert_allclose(g.predict({**g.build_input_dict(z, rot)}), img.detach().cpu().numpy(), atol=1e-4)
    assert float(g.weights[0].grad.abs().max()) == 0.0


def test_discriminator_loss_with_r1_double_backward():
    from confignet_amd.dnn_models.hologan_discriminator import HologanDiscriminator
    from confignet_amd.losses import compute_discriminator_loss
    rng = np.random.default_rng(7)
    res, n = 64, 3
    d = HologanDiscriminator((res, res), 5, 512, 3, 48, True, rng=rng)
    randomize(d, 2)
    real = rng.uniform(-1, 1, size=(n, res, res, 3))
    fake = rng.uniform(-1, 1, size=(n, res, res, 3))
    out = d(real)
    wr = w64(d)
    ref_out = R.discriminator_forward(wr, t64(real))
    assert list(out.keys()) == list(ref_out.keys())
    for k in out:
        close(out[k], ref_out[k], what=k)
    d.zero_grad()
    losses = compute_discriminator_loss(d, d.to_device(real), d.to_device(fake))
    torch.autograd.backward(losses["loss_sum"], inputs=d.trainable_weights)
    ref_losses = S.discriminator_loss(wr, t64(real), t64(fake))
    assert list(losses.keys()) == list(ref_losses.keys())
    for k in losses:
        close(losses[k], ref_losses[k], what=k)
    close_grads(d, None, "discriminator (R1)",
                recompute=lambda: S.grads_of(S.discriminator_loss(wr, t64(real), t64(fake))["loss_sum"], wr))
    # the literal reverse-over-reverse formulation (second-order tape on composite ops) agrees as well
    g_jvp = [p.grad.detach().clone() for p in d.weights]
    d.zero_grad()
    losses_t = compute_discriminator_loss(d, d.to_device(real), d.to_device(fake), second_order_tape=True)
    torch.autograd.backward(losses_t["loss_sum"], inputs=d.trainable_weights)
    for k in losses:
        close(losses_t[k], ref_losses[k], what="tape " + k)
    for a, p in zip(g_jvp, d.weights):
        rel = float((a - p.grad).norm() / (p.grad.norm() + 1e-30))
        assert rel < 2e-2, "JVP-R1 vs tape-R1 gradient: rel-L2 %.3e" % rel
    # the stacked tangent pass (all heads at once, HologanDiscriminator.tangent_all: the default above) against the six
    # head-by-head passes: the same arithmetic up to summation order
    from confignet_amd import losses as L
    d.zero_grad()
    prev, L.BATCHED_TANGENT = L.BATCHED_TANGENT, False
    try:
        losses_h = compute_discriminator_loss(d, d.to_device(real), d.to_device(fake))
    finally:
        L.BATCHED_TANGENT = prev
    torch.autograd.backward(losses_h["loss_sum"], inputs=d.trainable_weights)
    for k in losses:
        close(losses_h[k], losses[k].detach().cpu().double(), tol=1e-5, what="head-by-head " + k)
    for i, (a, p) in enumerate(zip(g_jvp, d.weights)):
        rel = float((a - p.grad).norm() / (p.grad.norm() + 1e-30))
        assert rel < 1e-4, "stacked vs head-by-head tangent pass, weight %d: rel-L2 %.3e" % (i, rel)


def test_latent_regressor_and_latent_discriminator():
    from confignet_amd.dnn_models.building_blocks import MLPSimple
    from confignet_amd.dnn_models.hologan_discriminator import HologanLatentRegressor
    from confignet_amd.losses import compute_latent_discriminator_loss
    rng = np.random.default_rng(8)
    lr = HologanLatentRegressor(43, (64, 64), 5, 512, 3, 48, True, rng=rng)
    randomize(lr, 3)
    img = rng.uniform(-1, 1, size=(2, 64, 64, 3))
    wr = w64(lr)
    out = lr(img)
    ref = R.latent_regressor_forward(wr, t64(img))
    close(out, ref, what="latent regressor")
    lr.zero_grad()
    torch.autograd.backward((out ** 2).sum(), inputs=lr.trainable_weights)
    close_grads(lr, None, "latent regressor",
                recompute=lambda: S.grads_of((R.latent_regressor_forward(wr, t64(img)) ** 2).sum(), wr))

    ld = MLPSimple(4, 43, 43, 1, rng=rng)
    randomize(ld, 4)
    a, b = rng.normal(size=(16, 43)), rng.normal(size=(16, 43))
    wr = w64(ld)
    ld.zero_grad()
    losses = compute_latent_discriminator_loss(ld, ld.to_device(a), ld.to_device(b))
    torch.autograd.backward(losses["loss_sum"], inputs=ld.trainable_weights)
    ref_losses = S.latent_discriminator_loss(wr, t64(a), t64(b))
    for k in losses:
        close(losses[k], ref_losses[k], what="latent D " + k)
    close_grads(ld, None, "latent discriminator (R1)",
                recompute=lambda: S.grads_of(S.latent_discriminator_loss(wr, t64(a), t64(b))["loss_sum"], wr))


@pytest.mark.parametrize("model_type", ["imagenet", "VGGFace"])
def test_perceptual_loss(model_type):
    from confignet_amd.perceptual_loss import PerceptualLoss
    rng = np.random.default_rng(9)
    pl = PerceptualLoss((64, 64, 3), model_type)
    gt = rng.uniform(-1, 1, size=(2, 64, 64, 3))
    gen = rng.uniform(-1, 1, size=(2, 64, 64, 3))
    gen_t = torch.tensor(gen, device="cuda", dtype=torch.float32, requires_grad=True)
    loss = pl.loss(gt, gen_t)
    (g,) = torch.autograd.grad(loss, gen_t)
    vw = w64(pl._pretrained_dnn_activations, grad=False)
    gen_r = t64(gen).requires_grad_(True)
    ref = R.perceptual_loss(vw, t64(gt), gen_r, model_type)
    (gr,) = torch.autograd.grad(ref, gen_r)
    close(loss, ref, tol=1e-3, what="perceptual loss", rel=True)
    rel = float((g.detach().cpu().double() - gr).norm() / gr.norm())
    assert rel < 2e-2, "d perceptual / d image: rel-L2 %.3e" % rel     # ReLU / max-pool argmax flips, see close_grads


@pytest.mark.parametrize("model_type", ["imagenet", "VGGFace"])
def test_perceptual_loss_as_one_tape_node_equals_the_layerwise_tape(model_type):
    """VGGLossFn (the stack + its four terms as one tape node, cn_tap_bwd at the taps) against one tape node per layer and term:
    the same loss and the same image gradient up to summation order, for loss() and for loss_groups() (two sample groups with
    different cotangents), with cached and with recomputed target features."""
    from confignet_amd.perceptual_loss import PerceptualLoss
    rng = np.random.default_rng(19)
    pl = PerceptualLoss((64, 64, 3), model_type)
    gt = torch.tensor(rng.uniform(-1, 1, size=(5, 64, 64, 3)), device="cuda", dtype=torch.float32)
    gen = rng.uniform(-1, 1, size=(5, 64, 64, 3))
    out = {}
    # deterministic mode: the K-split atomics of the direct convolutions move the activations by ~1e-6 from run to run, and one
    # ReLU / arg-max decision taken the other way moves a sample's gradient by ~1e-3 (seen on this very input) -- with ordered
    # reductions both forms see the same activations and differ by their own summation order only
    prev = ops_mod().DETERMINISTIC
    ops_mod().set_deterministic(True)
    for fused in (False, True):
        pl.fused_tape = fused
        a = torch.tensor(gen, device="cuda", dtype=torch.float32, requires_grad=True)
        l1 = pl.loss(gt, a)
        (g1,) = torch.autograd.grad(l1 * 3.0, a)
        b = torch.tensor(gen, device="cuda", dtype=torch.float32, requires_grad=True)
        l2 = pl.loss(b, gt, cached=pl.features(gt))
        (g2,) = torch.autograd.grad(l2, b)
        c = torch.tensor(gen, device="cuda", dtype=torch.float32, requires_grad=True)
        lg = pl.loss_groups(c, pl.features(gt), (2, 3))
        (g3,) = torch.autograd.grad(lg[0] * 0.5 + lg[1] * 2.0, c)
        out[fused] = [t.detach().clone() for t in (l1, g1, l2, g2, lg, g3)]
    ops_mod().set_deterministic(prev)
    pl.fused_tape = True
    for k, (x, y) in enumerate(zip(out[False], out[True])):
        err = float((x.double() - y.double()).norm() / x.double().norm())
        print("one node vs layerwise, item %d: rel-L2 %.3e" % (k, err))
        assert err < 1e-5, "item %d: rel-L2 %.3e" % (k, err)


def test_real_encoder():
    from confignet_amd.dnn_models.real_encoder import RealEncoder
    rng = np.random.default_rng(10)
    enc = RealEncoder(43, (64, 64, 3), ((-30, 30), (-10, 10), (0, 0)), rng=rng)
    randomize(enc, 5, 0.05)
    img = rng.uniform(-1, 1, size=(2, 64, 64, 3))
    wr = w64(enc)
    emb, rot = enc(img)
    emb_r, rot_r = R.real_encoder_forward(wr, t64(img))
    close(emb, emb_r, what="encoder embedding", rel=True)
    close(rot, rot_r, what="encoder rotation")
    enc.zero_grad()
    torch.autograd.backward((emb ** 2).sum() + (rot ** 2).sum() * 10, inputs=enc.trainable_weights)
    def ref_grads():
        e, r = R.real_encoder_forward(wr, t64(img))
        return torch.autograd.grad((e ** 2).sum() + (r ** 2).sum() * 10, wr, allow_unused=True)
    close_grads(enc, None, "real encoder", recompute=ref_grads)
    e2, r2 = enc.predict(img.astype(np.float32))
    np.testing.assert_allclose(e2, emb.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("sink", [False, True])
def test_real_encoder_folded_tape_equals_the_composite_form(sink):
    """The taped ResNet-50 as one node on the FOLDED filters (ResNetTrunkFn: bias / residual / ReLU in the convolutions' epilogues,
    skip gradients in the data gradients' epilogues, every parameter gradient from cn_bn_fold_bwd) against the composite form
    (convolution + per-channel affine pass per layer, autograd's adds): same outputs, same gradients of every parameter and of
    the image up to summation order -- through autograd's own accumulation and through a gradient sink (backward_into_arenas)."""
    from confignet_amd import nn as cnn
    from confignet_amd.dnn_models.real_encoder import RealEncoder
    rng = np.random.default_rng(11)
    enc = RealEncoder(43, (64, 64, 3), ((-30, 30), (-10, 10), (0, 0)), rng=rng)
    randomize(enc, 7, 0.05)
    img_np = rng.uniform(-1, 1, size=(3, 64, 64, 3)).astype(np.float32)
    out = {}
    for folded in (False, True):
        enc.folded_tape = folded
        img = torch.tensor(img_np, device="cuda", requires_grad=True)
        emb, rot = enc(img)
        loss = (emb ** 2).sum() + (rot ** 2).sum() * 10
        if sink:
            (gi,) = cnn.backward_into_arenas(loss, [enc], extra=[img])
        else:
            enc.zero_grad()
            for p_ in enc.trainable_weights:
                p_.grad = None
            grads = torch.autograd.grad(loss, enc.trainable_weights + [img])
            gi = grads[-1]
            enc.grad_arena.zero_()
            base = enc.arena.data_ptr()
            for p_, g_ in zip(enc.trainable_weights, grads[:-1]):
                o = (p_.data_ptr() - base) // 4
                enc.grad_arena[o:o + p_.numel()] += g_.reshape(-1)
        torch.cuda.synchronize()
        out[folded] = (emb.detach().clone(), rot.detach().clone(), gi.detach().clone(), enc.grad_arena.clone())
    enc.folded_tape = True
    a, b = out[False], out[True]
    for k, what in enumerate(("embedding", "rotation", "d image", "gradient arena")):
        ref = a[k].double()
        err = float((b[k].double() - ref).norm() / ref.norm())
        print("folded tape vs composite (%s, sink=%s): rel-L2 %.3e" % (what, sink, err))
        assert err < 2e-4, "%s: rel-L2 %.3e" % (what, err)
    # per-tensor: every kernel / bias / gamma / beta gradient, not only the arena's norm (which the large filters dominate)
    base = enc.arena.data_ptr()
    worst = (0.0, None)
    for i in enc._trainable_idx:
        p_ = enc.weights[i]
        o = (p_.data_ptr() - base) // 4
        ra, rb = a[3][o:o + p_.numel()].double(), b[3][o:o + p_.numel()].double()
        e = float((rb - ra).norm() / (ra.norm() + 1e-30))
        worst = max(worst, (e, enc._entries[i][0]))
    print("worst per-tensor rel-L2 %.3e (%s)" % worst)
    assert worst[0] < 2e-3, "per-tensor gradient: %s rel-L2 %.3e" % (worst[1], worst[0])


def test_conv_dgrad_with_residual_in_the_epilogue():
    """cn_conv_dgrad_w_res == cn_conv_dgrad_w followed by an add (same bits where the launch carries the residual), 1x1 and 3x3."""
    from confignet_amd import ops
    from confignet_amd.ops import ConvSpec
    rng = np.random.default_rng(12)
    for xs, k, cout in (((4, 16, 16, 256), 1, 64), ((2, 32, 32, 64), 1, 256), ((2, 16, 16, 64), 3, 64)):
        spec = ConvSpec((k, k))
        g = spec.geom(xs, cout)
        w = torch.tensor(rng.normal(size=(k, k, xs[-1], cout)) * 0.05, device="cuda", dtype=torch.float32)
        gy = torch.tensor(rng.normal(size=(xs[0], xs[1], xs[2], cout)), device="cuda", dtype=torch.float32)
        res = torch.tensor(rng.normal(size=xs), device="cuda", dtype=torch.float32)
        want = ops.conv_dgrad(gy, w, g) + res
        got = ops.conv_dgrad_res(gy, w, g, res)
        err = float((got - want).abs().max())
        assert err <= 1e-5 * float(want.abs().max()), "case %s k=%d: max-abs %.3e" % (xs, k, err)


def _make_model(cls, res, batch, seed=0):
    cfg = {"output_shape": (res, res, 3), "batch_size": batch, "facemodel_inputs": dict(FM)}
    np.random.seed(seed)
    m = cls(cfg, seed=seed)
    for i, net in enumerate(m.all_networks()):
        if net is not m.generator_smoothed:
            randomize(net, 100 + i, 0.05)
    m.generator_smoothed.copy_weights_from(m.generator)
    return m


def _oracle_weights(m):
    names = {"generator": m.generator, "discriminator": m.discriminator, "synth_discriminator": m.synth_discriminator,
             "latent_discriminator": m.latent_discriminator, "latent_regressor": m.latent_regressor,
             "synthetic_encoder": m.synthetic_encoder}
    if getattr(m, "encoder", None) is not None:
        names["real_encoder"] = m.encoder
    return {k: w64(v) for k, v in names.items()}


def _batch(m, res, n_synth, n_real, rng):
    params = [rng.normal(size=(n_synth, d[0])) for d in m.config["facemodel_inputs"].values()]
    rot = rng.uniform(-0.4, 0.4, size=(n_synth + n_real, 3))
    rot[:, 2] = 0
    imgs = rng.uniform(-1, 1, size=(n_synth + n_real, res, res, 3))
    masks = np.zeros((n_synth, res, res), np.uint8)
    masks[:, 20:30, 30:45] = 1
    return params, rot, imgs, masks


def test_first_stage_generator_step_and_adam():
    from confignet_amd import ConfigNetFirstStage, optim
    res, ns, nr = 128, 1, 1
    m = _make_model(ConfigNetFirstStage, res, ns + nr)
    rng = np.random.default_rng(11)
    params, rot, imgs, masks = _batch(m, res, ns, nr, rng)
    z_real = rng.normal(size=(nr, m.config["latent_dim"]))
    W = _oracle_weights(m)
    vgg_w = w64(m.perceptual_loss._pretrained_dnn_activations, grad=False)
    dev = m._dev
    nets = [m.generator, m.latent_regressor, m.synthetic_encoder]
    for n in nets:
        n.zero_grad()
    from confignet_amd.confignet_first_stage import frozen
    opt = optim.Adam(**m.config["optimizer"])
    with frozen(m.discriminator, m.synth_discriminator, m.latent_discriminator):
        losses = m._generator_loss([dev(p) for p in params], dev(rot[:ns]), dev(imgs[:ns]),
                                   torch.as_tensor(masks).cuda(), dev(z_real), dev(rot[ns:]))
        with ops_mod().branch_log() as branch_log:
            torch.autograd.backward(losses["loss_sum"], inputs=[p for n in nets for p in n.trainable_weights])
    ref, _ = S.first_stage_generator_loss(W, m.config, [t64(p) for p in params], t64(rot[:ns]), t64(imgs[:ns]),
                                          torch.as_tensor(masks), t64(z_real), t64(rot[ns:]), vgg_w)
    assert list(losses.keys()) == list(ref.keys())
    for k in losses:
        close(losses[k], ref[k], what="G step " + k)
    allw = W["generator"] + W["latent_regressor"] + W["synthetic_encoder"]
    grads = S.grads_of(ref["loss_sum"], allw)
    ng, nl = len(W["generator"]), len(W["latent_regressor"])

    def ref_grads():
        r, _ = S.first_stage_generator_loss(W, m.config, [t64(p) for p in params], t64(rot[:ns]), t64(imgs[:ns]),
                                            torch.as_tensor(masks), t64(z_real), t64(rot[ns:]), vgg_w)
        return S.grads_of(r["loss_sum"], allw)
    # every LeakyReLU / ReLU / max-pool decision of the product's backward pass forced in the oracle: what is left is summation
    # error, held at the single networks' 5e-3 in default AND deterministic mode (rounds 1-3 bounded this chain at 3e-2 with a
    # greedy fit of candidate flips: which near-zero elements a run flips moved the deviation between 5e-3 and 2.2e-2)
    check_grads_forced([(m.generator, "G step: generator", slice(0, ng)), (m.latent_regressor, "G step: latent regressor", slice(ng, ng + nl)),
                        (m.synthetic_encoder, "G step: synthetic encoder", slice(ng + nl, None))], ref_grads, branch_log, tol=1e-3)
    # (measured over 8 runs, default and deterministic mode: 2.2e-5 .. 3.3e-5, 6-13 of the decisions of 61 activations forced)
    # Keras Adam (shared counter) + EMA on the arenas vs the oracle
    # With beta_1 = 0 the first Keras-Adam step is lr*sign(g): entries whose gradient is at noise level may
    # take the other sign than the float64 oracle, so the update is compared where |g| is significant.
    old = {id(net): [p.detach().clone() for p in net.weights] for net in nets}
    ropt = O.KerasAdam(**m.config["optimizer"])
    before = [w.detach().clone() for w in allw]
    ropt.apply_gradients(list(zip(grads, allw)))
    opt.apply_gradients(nets)
    k = 0
    for net in nets:
        for p, p_old in zip(net.weights, old[id(net)]):
            g_ref, step_ref = grads[k], (allw[k].detach() - before[k])
            k += 1
            sig = g_ref.abs() > 0.1 * g_ref.abs().max()
            if not bool(sig.any()):
                continue
            step = (p.detach() - p_old).cpu().double()
            assert float((step - step_ref)[sig].abs().max()) < 2e-6, "adam step mismatch"
            assert float(step.abs().max()) <= 4e-4 * 1.001
    for ws in (W["generator"],):
        pass
    sm = [w.detach().clone() for w in w64(m.generator_smoothed, grad=False)]
    S.ema_update(sm, w64(m.generator, grad=False))
    m.update_smoothed_weights()
    for p, r in zip(m.generator_smoothed.weights, sm):
        close(p, r, tol=1e-6, what="EMA weight")


@pytest.mark.parametrize("stacked", [False, True])
def test_second_stage_generator_step(stacked):
    """stacked: ConfigNet.merge_generator_passes (CN_G_MERGE=1; off by default -- it measured slower end to end): one stacked
    generator / VGG pass for the synthetic and the real half, held to the same oracle comparison as the two-pass form."""
    from confignet_amd import ConfigNet
    from confignet_amd.confignet_first_stage import frozen
    res, ns, nr = 128, 1, 1
    m = _make_model(ConfigNet, res, ns + nr, seed=1)
    m.merge_generator_passes = stacked
    rng = np.random.default_rng(12)
    params, rot, imgs, masks = _batch(m, res, ns, nr, rng)
    W = _oracle_weights(m)
    vgg_w = w64(m.perceptual_loss._pretrained_dnn_activations, grad=False)
    dev = m._dev
    nets = [m.generator, m.latent_regressor, m.synthetic_encoder, m.encoder]
    for n in nets:
        n.zero_grad()
    with frozen(m.discriminator, m.synth_discriminator, m.latent_discriminator):
        losses = m._generator_loss([dev(p) for p in params], dev(rot[:ns]), dev(imgs[:ns]),
                                   torch.as_tensor(masks).cuda(), dev(imgs[ns:]))
        with ops_mod().branch_log() as branch_log:
            torch.autograd.backward(losses["loss_sum"], inputs=[p for n in nets for p in n.trainable_weights])
    ref, _ = S.second_stage_generator_loss(W, m.config, [t64(p) for p in params], t64(rot[:ns]), t64(imgs[:ns]),
                                           torch.as_tensor(masks), t64(imgs[ns:]), vgg_w)
    assert list(losses.keys()) == list(ref.keys())
    for k in losses:
        close(losses[k], ref[k], what="stage-2 G step " + k)
    allw = W["generator"] + W["latent_regressor"] + W["synthetic_encoder"] + W["real_encoder"]
    ng, nl, ne = len(W["generator"]), len(W["latent_regressor"]), len(W["synthetic_encoder"])

    def ref_grads():
        r, _ = S.second_stage_generator_loss(W, m.config, [t64(p) for p in params], t64(rot[:ns]), t64(imgs[:ns]),
                                             torch.as_tensor(masks), t64(imgs[ns:]), vgg_w)
        return torch.autograd.grad(r["loss_sum"], allw, allow_unused=True)
    # the deepest chain of the suite -- generator + VGG-19 + ResNet-50 + six discriminator heads, millions of ReLU / LeakyReLU /
    # max-pool decisions.  Rounds 1-3 bounded it at 8e-2, then 1.5e-1: which near-zero elements the GPU decides differently from
    # float64 changed from run to run (1.2e-2 .. 1.04e-1).  With the product's own decisions forced in the oracle
    # (check_grads_forced) the comparison is deterministic and the chain is held at the single networks' 5e-3.
    check_grads_forced([(m.generator, "stage-2: generator", slice(0, ng)), (m.latent_regressor, "stage-2: latent regressor", slice(ng, ng + nl)),
                        (m.synthetic_encoder, "stage-2: synthetic encoder", slice(ng + nl, ng + nl + ne)),
                        (m.encoder, "stage-2: real encoder", slice(ng + nl + ne, None))], ref_grads, branch_log, tol=2e-3)
    # (measured over 8 runs, default and deterministic mode: 5.1e-4 .. 7.9e-4; ~9 000 of the millions of decisions of 127
    # activations / pools taken the product's way, none further than 3.6e-4 of the tensor's mean magnitude from zero)


def test_full_iteration_runs_and_api(tmp_path):
    """One whole second-stage iteration through the reference-shaped API on a synthetic dataset;
    save/load round trip in the reference's npz/json layout; generate_images uint8 contract."""
    from confignet_amd import ConfigNet, SyntheticFaceDataset, load_confignet, optim
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.confignet_utils import merge_configs
    np.random.seed(0)
    ds = SyntheticFaceDataset(16, 128, seed=3)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "output_shape": (128, 128, 3)})
    ds.process_metadata(cfg, True)
    m = ConfigNet(cfg, seed=0)
    assert m.config["latent_dim"] == 145
    m.setup_training(None, ds, 0, real_training_set=ds)
    dopt, gopt = optim.Adam(**m.config["optimizer"]), optim.Adam(**m.config["optimizer"])
    before = m.generator.arena.clone()
    for _ in range(2):
        d, sd, ld, g = m.training_iteration(ds, ds, dopt, gopt)
    assert dopt.iterations == 6 and gopt.iterations == 2            # shared-counter rule (R10)
    for losses in (d, sd, ld, g):
        assert np.isfinite(float(losses["loss_sum"]))
    assert set(d.keys()) == {"GAN_loss_real_%d" % i for i in range(6)} | {"GAN_loss_fake_%d" % i for i in range(6)} | \
        {"gp_loss_%d" % i for i in range(6)} | {"loss_sum"}
    assert not torch.equal(before, m.generator.arena)
    imgs = m.generate_images(m.sample_latent_vector(3), m.sample_rotations(3))
    assert imgs.shape == (3, 128, 128, 3) and imgs.dtype == np.uint8
    emb, rot = m.encode_images(ds.imgs[:2])
    assert emb.shape == (2, 145) and rot.shape == (2, 3)
    lat2 = m.set_facemodel_param_in_latents(emb, "blendshape_values", np.zeros(62, np.float32))
    assert lat2.shape == emb.shape and np.array_equal(np.nonzero((lat2 != emb).any(0))[0], np.arange(7, 37))
    m.save(str(tmp_path), "model")
    m2 = load_confignet(str(tmp_path / "model.json"))
    assert type(m2).__name__ == "ConfigNet"
    # fp32 atomics in the statistics kernels make the last bit run-dependent: uint8 images agree to 1 level
    assert np.abs(m2.generate_images(emb, rot).astype(int) - m.generate_images(emb, rot).astype(int)).max() <= 1
    e3, r3 = m.fine_tune_on_img(ds.imgs[:1], n_iters=2)
    assert e3.shape == (1, 145) and np.isfinite(e3).all()
    # only the expression slice differs from a stale pre/post return after the steps
    assert m.generator_fine_tuned is not None
    # cached target activations (default) == recomputing them every step as the reference does
    first = {}
    for cache in (False, True):
        m.cache_target_features = cache
        m.fine_tune_on_img(ds.imgs[:1], n_iters=1)
        first[cache] = {k: float(v) for k, v in m.last_fine_tune_losses.items()}
    for k, v in first[False].items():          # (fp32 atomics in the statistics kernels: runs agree to ~1e-4, not bitwise)
        assert abs(first[True][k] - v) <= 1e-3 * max(1.0, abs(v)), (k, first[True][k], v)


def test_latent_gan_step():
    from confignet_amd import LatentGAN, optim
    np.random.seed(0)
    gan = LatentGAN({"latent_dim": 145, "batch_size": 256}, seed=0)
    opt = optim.Adam(**gan.config["optimizer"])
    emb = np.random.normal(size=(1000, 145)).astype(np.float32)
    # parity of the discriminator loss incl. R1 with the oracle
    real, fake = emb[:64], emb[64:128] * 0.5
    from oracle.ref_steps import latent_discriminator_loss, grads_of
    wr = w64(gan.discriminator)
    gan.discriminator.zero_grad()
    losses = gan._discriminator_loss(gan.discriminator.to_device(real), gan.discriminator.to_device(fake))
    torch.autograd.backward(losses["loss_sum"], inputs=gan.discriminator.trainable_weights)
    ref = latent_discriminator_loss(wr, t64(real), t64(fake))
    for k in losses:
        close(losses[k], ref[k], what="LatentGAN D " + k)
    close_grads(gan.discriminator, None, "LatentGAN D",
                recompute=lambda: grads_of(latent_discriminator_loss(wr, t64(real), t64(fake))["loss_sum"], wr))
    for _ in range(2):
        d = gan.discriminator_training_step(emb, opt)
        g = gan.generator_training_step(opt)
        gan.update_smoothed_weights()
    assert np.isfinite(float(d["loss_sum"])) and np.isfinite(float(g["loss_sum"]))
    assert gan.generate_latents(5).shape == (5, 145)


def test_fused_norm_functions_match_composites_and_oracle():
    """AdaInFn / DiscrTailFn (fused, first order) == the twice-differentiable composites == the oracle."""
    from confignet_amd import functional as F
    rng = np.random.default_rng(21)
    x = rng.normal(size=(3, 8, 8, 8, 32)) * 2 + 0.3
    sb = rng.normal(size=(3, 64))
    cot = rng.normal(size=x.shape)

    def run(fn):
        xt = torch.tensor(x, device="cuda", dtype=torch.float32, requires_grad=True)
        st = torch.tensor(sb, device="cuda", dtype=torch.float32, requires_grad=True)
        y = fn(xt, st)
        gx, gs = torch.autograd.grad((y * torch.tensor(cot, device="cuda", dtype=torch.float32)).sum(), [xt, st])
        return y, gx, gs
    yf, gxf, gsf = run(F.adain)
    yc, gxc, gsc = run(F.adain_composite)
    xr, sr = t64(x).requires_grad_(True), t64(sb).requires_grad_(True)
    mu = xr.mean(dim=(1, 2, 3), keepdim=True)
    var = ((xr - mu) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    yr = (xr - mu) * torch.rsqrt(var + 1e-3) * (sr[:, :32].reshape(3, 1, 1, 1, 32) + 1) + sr[:, 32:].reshape(3, 1, 1, 1, 32)
    gxr, gsr = torch.autograd.grad((yr * t64(cot)).sum(), [xr, sr])
    for got, ref, what in ((yf, yr, "adain y"), (gxf, gxr, "adain gx"), (gsf, gsr, "adain gsb"),
                           (yc, yr, "composite y"), (gxc, gxr, "composite gx"), (gsc, gsr, "composite gsb")):
        close(got, ref, tol=2e-4, what=what, rel=True)

    # DiscrBlock tail: style + LeakyReLU + instance norm
    x = rng.normal(size=(2, 16, 16, 48))
    gamma, beta = rng.normal(size=48) * 0.2 + 1, rng.normal(size=48) * 0.2
    cy, cs = rng.normal(size=x.shape), rng.normal(size=(2, 96))
    xt = torch.tensor(x, device="cuda", dtype=torch.float32, requires_grad=True)
    gt = torch.tensor(gamma, device="cuda", dtype=torch.float32, requires_grad=True)
    bt = torch.tensor(beta, device="cuda", dtype=torch.float32, requires_grad=True)
    y, st = F.DiscrTailFn.apply(xt, gt, bt, True, 0.3)[:2]
    loss = (y * torch.tensor(cy, device="cuda", dtype=torch.float32)).sum() + (st * torch.tensor(cs, device="cuda", dtype=torch.float32)).sum()
    g = torch.autograd.grad(loss, [xt, gt, bt])
    xr, gr, br = t64(x).requires_grad_(True), t64(gamma).requires_grad_(True), t64(beta).requires_grad_(True)
    mu, sd = O.layer_style(xr)
    st_r = torch.cat([mu, sd], dim=-1).reshape(2, -1)
    y_r = O.instance_norm(O.leaky_relu(xr, 0.3), gr, br)
    g_r = torch.autograd.grad((y_r * t64(cy)).sum() + (st_r * t64(cs)).sum(), [xr, gr, br])
    close(y, y_r, tol=2e-4, what="tail y", rel=True)
    close(st, st_r, tol=2e-4, what="tail style", rel=True)
    for a, b, what in zip(g, g_r, ("tail gx", "tail ggamma", "tail gbeta")):
        close(a, b, tol=5e-4, what=what, rel=True)


def test_hip_graph_steps_match_eager():
    """Each step's device half captured into a HIP graph (forward, R1 tangent pass, backward, Adam with the
    device-side lr_t scalar) replays to the same losses / weight updates as eager dispatch FROM THE SAME STATE:
    two graph iterations (eager warm-up + capture, then a pure replay), snapshot, one more replayed iteration,
    restore the snapshot and run the same iteration eagerly."""
    import os
    from confignet_amd import ConfigNet, SyntheticFaceDataset, optim, parallel
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.confignet_utils import merge_configs
    forced_dp = os.environ.get("CN_FORCE_DP", "0") == "1"      # set by test_data_parallel_graph_path_single_rank
    if forced_dp:
        parallel.init_from_env()
        assert parallel.active()
    ds = SyntheticFaceDataset(16, 128, seed=3)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "output_shape": (128, 128, 3)})
    ds.process_metadata(cfg, True)
    np.random.seed(5)
    m = ConfigNet(cfg, seed=0)
    m.use_graphs = True
    m.setup_training(None, ds, 0, real_training_set=ds)
    dopt, gopt = optim.Adam(**m.config["optimizer"]), optim.Adam(**m.config["optimizer"])
    for _ in range(3):
        m.training_iteration(ds, ds, dopt, gopt)
    assert len(m._graphs) == 4 and all(g.graph is not None for g in m._graphs.values())
    assert all(bool(g.tail) == forced_dp for g in m._graphs.values())     # DP: all-reduce + Adam outside the graph
    assert dopt.iterations == 9 and gopt.iterations == 3
    nets = m.all_networks()

    def snapshot():
        torch.cuda.synchronize()
        return {"w": [n.arena.clone() for n in nets], "nt": [[w.clone() for w in n.weights if not w.requires_grad] for n in nets],
                "opt": [{k: (a.clone(), b.clone()) for k, (a, b) in o._state.items()} for o in (dopt, gopt)],
                "it": (dopt.iterations, gopt.iterations), "rng": np.random.get_state()}

    def restore(s):
        for n, a in zip(nets, s["w"]):
            n.arena.copy_(a)
            n.mark_updated()
        for o, st in zip((dopt, gopt), s["opt"]):
            for k, (a, b) in st.items():
                o._state[k][0].copy_(a)
                o._state[k][1].copy_(b)
        dopt.iterations, gopt.iterations = s["it"]
        np.random.set_state(s["rng"])

    snap = snapshot()
    out_g = [{k: float(v) for k, v in d.items()} for d in m.training_iteration(ds, ds, dopt, gopt)]
    w_g = [n.arena.clone() for n in nets]
    restore(snap)
    m.use_graphs = False
    out_e = [{k: float(v) for k, v in d.items()} for d in m.training_iteration(ds, ds, dopt, gopt)]
    for dg, de in zip(out_g, out_e):
        assert dg.keys() == de.keys()
        for k in dg:
            assert abs(dg[k] - de[k]) <= 3e-3 * max(1.0, abs(de[k])), (k, dg[k], de[k])   # fp32-atomics noise
    lr = 4e-4
    for n, a, w0 in zip(nets, w_g, snap["w"]):
        diff = (n.arena - a).abs()
        moved = (a - w0).abs().max()
        # identical up to fp32-atomics noise: only gradients at noise level may take the other sign
        # |Adam step| <= lr_t/sqrt(1-beta_2) = 3.17 lr when beta_1 = 0; two opposite-sign steps differ by twice that
        # ... and on average the two runs' updates differ by a small fraction of the update itself
        upd = float((a - w0).abs().mean())
        assert float(diff.max()) <= 6.5 * lr and float(diff.mean()) <= 0.05 * upd + 1e-7, (float(diff.max()), float(diff.mean()), upd)
        if n is not m.generator_smoothed:
            assert float(moved) > 0 or n.n_trainable == 0


def test_data_parallel_graph_path_single_rank():
    """The multi-rank dispatch (graphs hold forward+backward; RCCL all-reduce of the gradient arenas and Adam are
    issued eagerly after each replay, also from the side streams of the concurrent discriminator phase) exercised on
    one GPU: CN_FORCE_DP=1 creates a 1-rank RCCL process group and routes through exactly that code."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, CN_FORCE_DP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_nets_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-k", "test_hip_graph_steps_match_eager"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_independent_streams_run_side_by_side():
    """graphs.independent_streams picks replay streams by measurement: two busy-wait launches on any two of them must
    overlap (they do not when two streams share a hardware queue)."""
    import time
    from confignet_amd import graphs
    from confignet_amd._lib import lib
    sts = graphs.independent_streams(3)
    assert len(sts) == 3 and len({s.cuda_stream for s in sts}) == 3

    def spins(a, b):
        best = 1e9
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            assert lib.cn_spin(50000, a.cuda_stream) == 0 and lib.cn_spin(50000, b.cuda_stream) == 0   # 0.5 ms each
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best
    serial = spins(sts[0], sts[0])
    assert 0.9e-3 < serial < 5e-3, serial
    for i in range(3):
        for j in range(i + 1, 3):
            assert spins(sts[i], sts[j]) < 0.8 * serial, (i, j, spins(sts[i], sts[j]), serial)


def test_train_loops_run_with_graph_dispatch(tmp_path, capsys):
    """ConfigNetFirstStage.train / ConfigNet.train as train_confignet.py drives them (l.57-71): HIP-graph dispatch by
    default, loss logs grow by one entry per iteration, periodic checkpoints land in output_dir/checkpoints."""
    import os
    from confignet_amd import ConfigNet, ConfigNetFirstStage, SyntheticFaceDataset
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.confignet_utils import merge_configs
    np.random.seed(11)
    ds = SyntheticFaceDataset(12, 128, seed=4)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "output_shape": (128, 128, 3), "metrics_checkpoint_period": 3})
    ds.process_metadata(cfg, True)
    first = ConfigNetFirstStage(cfg, seed=0)
    first.train(ds, ds, str(tmp_path / "first"), None, n_steps=5)
    assert first.use_graphs and first.get_training_step_number() == 4      # (the reference counts len(log) - 1)
    assert all(np.isfinite(v).all() for v in first.g_losses.values()) and len(first.g_losses["loss_sum"]) == 5
    assert os.path.exists(tmp_path / "first" / "checkpoints" / "000003.json")
    second = ConfigNet(cfg, seed=0)
    ConfigNetFirstStage.set_weights(second, first.get_weights())
    second.train(ds, ds, None, None, str(tmp_path / "second"), None, n_steps=4)
    assert second.get_training_step_number() == 3 and np.isfinite(second.g_losses["loss_sum"]).all()
    assert "[D loss:" in capsys.readouterr().out


@pytest.mark.parametrize("res,batch,graphs", [(512, 2, False), (128, 3, True), (256, 5, True)])
def test_other_resolutions_and_odd_batches(res, batch, graphs):
    """BASELINE's other shapes: 512x512 (one more upsampling / resampling level everywhere), batches that do not split
    evenly into the synthetic and real halves (confignet_second_stage.py:150-151: n_synth = B // 2)."""
    from confignet_amd import ConfigNet, SyntheticFaceDataset, optim
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.confignet_utils import merge_configs
    np.random.seed(0)
    ds = SyntheticFaceDataset(6, res, seed=1)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": batch, "output_shape": (res, res, 3)})
    ds.process_metadata(cfg, True)
    m = ConfigNet(cfg, seed=0)
    m.setup_training(None, ds, 0, real_training_set=ds)
    d, g = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
    m.use_graphs = graphs
    for _ in range(4 if graphs else 2):
        out = m.training_iteration(ds, ds, d, g)
    assert all(np.isfinite(float(o["loss_sum"])) for o in out)
    imgs = m.generate_images(m.sample_latent_vector(2), m.sample_rotations(2))
    assert imgs.shape == (2, res, res, 3) and imgs.dtype == np.uint8
    m1 = ConfigNet(merge_configs(cfg, {"batch_size": 1}), seed=0)
    m1.setup_training(None, ds, 0, real_training_set=ds)
    with pytest.raises(AssertionError, match="batch_size >= 2"):
        m1.generator_training_step(ds, ds, g)
