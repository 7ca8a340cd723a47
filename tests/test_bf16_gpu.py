"""bf16 compute path (BASELINE.json configs[2]: bf16 activations / filter copies on v_mfma_f32_32x32x16_bf16 with fp32
accumulation; fp32 master weights, statistics, losses, gradients of parameters and optimizer state).

Tolerances are stated per test.  A bf16 value carries 8 significant bits (relative spacing 2^-8 = 3.9e-3, rounding error
<= 2^-9 = 1.95e-3):
  * raw convolutions are compared with the float64 oracle evaluated on the bf16-ROUNDED operands, so only the fp32
    accumulation order and the final rounding of a bf16 output remain: |err| <= 2^-8 * |y| + fp32 noise (outputs);
    the fp32 filter gradient has no output rounding: 2e-4 of its scale, like the fp32 family;
  * whole networks / losses are compared with the fp32-exact oracle on the unrounded weights: every layer rounds its
    output once, so the error grows like sqrt(depth) * 2^-9 relative: 3e-2 of the output scale is asserted for the
    generator image (relative L2; single pixels up to 0.2 of the tanh range) and the loss scalars
    (scripts/dev/bf16_diag.py prints the layer-by-layer growth)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_nets as R
from oracle import ref_ops as O
from oracle import ref_steps as S


@pytest.fixture(autouse=True)
def _bf16_mode():
    from confignet_amd import ops
    ops.set_activation_dtype("bf16")
    yield
    ops.set_activation_dtype("f32")


def bf16_round(a):
    return torch.tensor(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float64)


def dev_bf16(a):
    return torch.tensor(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).cuda().contiguous()


def dev(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).cuda().contiguous()


def t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


# bounds of the comparisons against the oracle that rounds what the product rounds (oracle.ref_ops.bf16_storage).  Measured
# (round 5): discriminator heads 6.2e-3 of their scale against 2.4e-2 for the plain fp32 oracle -- four times sharper, so a
# systematic 3 % error of one block no longer passes; generator image rel-L2 9.2e-3 against 1.4e-2 -- only 1.5 times: 15
# storage points deep, a value on a rounding boundary in one layer moves a whole receptive field in the next, and the emulation
# does not restate the fp32 coefficient algebra of AdaIn bit for bit.  The generator bound is therefore kept as a record, not as
# a sharp statement.
BF16_EMU_GEN_REL = 2e-2
BF16_EMU_DISCR_ERR = 1.5e-2

BF16_CONV_CASES = [
    # (x shape, kernel, cout, stride, up, act, slope)
    ((2, 16, 16, 64), (4, 4), 32, 1, 1, 1, 0.3),        # k4 + folded upsample, 128x32 tile
    ((2, 16, 16, 512), (4, 4), 256, 1, 0, 1, 0.3),      # map_2d_0, 64x64 tiles
    ((2, 4, 4, 4, 512), (3, 3, 3), 256, 1, 1, 1, 0.3),  # map_3d_0 with folded upsample
    ((1, 8, 8, 8, 64), (3, 3, 3), 64, 1, 0, 1, 0.3),    # map_3d_post
    ((3, 16, 16, 1024), (1, 1), 512, 1, 0, 1, 0.2),     # projection conv
    ((2, 32, 32, 48), (3, 3), 96, 2, 0, 0, 0.0),        # D block 1: cin = 48 (a 32-deep stage half empty), stride 2 -> parity-ordered dgrad
    ((2, 17, 13, 48), (3, 3), 96, 2, 0, 0, 0.0),        # ragged odd extents (dgrad without parity order)
    ((16, 32, 32, 96), (3, 3), 192, 2, 0, 0, 0.0),      # D block 2, 128x96 tiles
    ((4, 64, 64, 64), (3, 3), 64, 1, 0, 2, 0.0),        # VGG conv1_2: 128x64 tiles + relu
    ((16, 32, 32, 128), (3, 3), 256, 1, 0, 2, 0.0),     # 128x128 tiles
    ((4, 8, 8, 256), (3, 3), 512, 1, 0, 2, 0.0),        # VGG block4 shape, 64x64 tiles
    ((1, 8, 8, 128), (1, 1), 512, 2, 0, 0, 0.0),        # ResNet strided 1x1
    ((2, 9, 9, 72), (3, 3), 40, 1, 0, 0, 0.0),          # channel counts that are multiples of 8 only
]


@pytest.mark.parametrize("case", BF16_CONV_CASES, ids=[str(i) for i in range(len(BF16_CONV_CASES))])
def test_bf16_conv_fwd_dgrad_wgrad(case):
    from confignet_amd import ops
    xs, k, cout, stride, up, act, slope = case
    rng = np.random.default_rng(abs(hash(case)) % 2 ** 31)
    cin = xs[-1]
    x = rng.normal(size=xs)
    w = rng.normal(size=(*k, cin, cout)) / math.sqrt(np.prod(k) * cin)
    b = rng.normal(size=cout)
    g = ops.ConvSpec(k, stride=stride, up=up).geom(xs, cout)
    wd = dev(w)
    y = ops.conv_fwd(dev_bf16(x), wd, dev(b), g, act, slope)
    assert y.dtype == torch.bfloat16
    xr, wr = bf16_round(x).requires_grad_(True), bf16_round(w).requires_grad_(True)
    xu = O.upsample2(xr) if up else xr
    xu.retain_grad()

    def conv(xx, bias, a):
        yy = O.conv_same(xx, wr, bias, stride=stride)
        return O.leaky_relu(yy, slope) if a == 1 else torch.relu(yy) if a == 2 else yy
    yr = conv(xu, t64(b), act).detach()
    err = (y.double().cpu() - yr).abs()
    assert float((err - 2.0 ** -8 * yr.abs()).max()) <= 2e-4 * max(1.0, float(yr.abs().max())), float(err.max())
    yr0 = conv(xu, None, 0)
    gy = rng.normal(size=tuple(yr0.shape))
    gyr = bf16_round(gy)
    (yr0 * gyr).sum().backward()
    gu = ops.conv_dgrad(dev_bf16(gy), wd, g)
    assert gu.dtype == torch.bfloat16
    err = (gu.double().cpu() - xu.grad).abs()
    assert float((err - 2.0 ** -8 * xu.grad.abs()).max()) <= 2e-4 * max(1.0, float(xu.grad.abs().max())), float(err.max())
    if up:
        gx = ops.sumpool2(gu)                    # sums 4 / 8 bf16 children in fp32, rounds once
        ref = xr.grad
        assert gx.dtype == torch.bfloat16
        assert float((gx.double().cpu() - ref).abs().max()) <= 3 * 2.0 ** -8 * float(xu.grad.abs().max()) * (2 ** len(k)) ** 0.5
    gw = ops.conv_wgrad(dev_bf16(x), dev_bf16(gy), g, tuple(w.shape))
    assert gw.dtype == torch.float32
    assert float((gw.double().cpu() - wr.grad).abs().max()) <= 2e-4 * max(1.0, float(wr.grad.abs().max()))


@pytest.mark.parametrize("xs,cout,stride", [((2, 64, 64, 3), 48, 2), ((1, 9, 150, 3), 64, 1), ((1, 11, 301, 3), 20, 2)])
def test_first_layer_filter_gradient_with_a_bf16_output_gradient(xs, cout, stride):
    """K = 27 filter gradient (cn_conv_wgrad_c3): the 3-channel image stays fp32, the output gradient arrives in bf16."""
    from confignet_amd import ops
    rng = np.random.default_rng(cout + stride)
    x = rng.normal(size=xs)
    g = ops.ConvSpec((3, 3), stride=stride).geom(xs, cout)
    wr = torch.zeros(3, 3, 3, cout, dtype=torch.float64, requires_grad=True)
    y = O.conv_same(t64(x.astype(np.float32)), wr, None, stride=stride)
    gy = rng.normal(size=tuple(y.shape))
    (y * bf16_round(gy)).sum().backward()
    gw = ops.conv_wgrad(dev(x), dev_bf16(gy), g, (3, 3, 3, cout))
    assert gw.dtype == torch.float32
    assert float((gw.double().cpu() - wr.grad).abs().max()) <= 2e-4 * max(1.0, float(wr.grad.abs().max()))


@pytest.mark.parametrize("xs,cout,stride", [((2, 64, 64, 3), 48, 2), ((2, 37, 45, 3), 48, 2), ((2, 32, 32, 3), 64, 1)])
def test_first_layer_forward_and_image_gradient_with_mixed_storage(xs, cout, stride):
    """bf16 path: the 3x3 convolution of the fp32 image writes bf16 directly (cn_conv_fwd_dt), and the data gradient into the
    image reads the bf16 output gradient directly (cn_conv_dgrad_dt); both against the float64 oracle."""
    from confignet_amd import ops
    rng = np.random.default_rng(cout * stride)
    x, w, b = rng.normal(size=xs), rng.normal(size=(3, 3, 3, cout)) / math.sqrt(27), rng.normal(size=cout)
    g = ops.ConvSpec((3, 3), stride=stride).geom(xs, cout)
    ops.set_activation_dtype("bf16")
    try:
        y = ops.conv_fwd(dev(x), dev(w), dev(b), g, 1, 0.3)
        assert y.dtype == torch.bfloat16
        xr = t64(x.astype(np.float32)).requires_grad_(True)
        wr = t64(w.astype(np.float32))
        yr = O.leaky_relu(O.conv_same(xr, wr, t64(b.astype(np.float32)), stride=stride), 0.3).detach()
        err = (y.double().cpu() - yr).abs()
        assert float((err - 2.0 ** -8 * yr.abs()).max()) <= 2e-4 * max(1.0, float(yr.abs().max())), float(err.max())
        y0 = O.conv_same(xr, wr, None, stride=stride)
        gy = rng.normal(size=tuple(y0.shape))
        (y0 * bf16_round(gy)).sum().backward()
        gx = ops.conv_dgrad(dev_bf16(gy), dev(w), g)
        assert gx.dtype == torch.float32                       # 3-channel tensors stay fp32
        assert float((gx.double().cpu() - xr.grad).abs().max()) <= 2e-4 * max(1.0, float(xr.grad.abs().max()))
    finally:
        ops.set_activation_dtype("f32")


def test_bf16_elementwise_family_matches_fp32_math_on_the_same_bits():
    """Statistics / affine / activation / pooling kernels in bf16 storage: identical arithmetic to the fp32 kernels applied to
    the bf16 values (fp32 accumulate), outputs rounded once."""
    from confignet_amd import ops
    rng = np.random.default_rng(5)
    x = rng.normal(size=(3, 9, 11, 40)) * 2 + 0.5
    y = rng.normal(size=(3, 9, 11, 40))
    xb, yb = dev_bf16(x), dev_bf16(y)
    xr, yr = bf16_round(x), bf16_round(y)
    s1, s2 = ops.nc_reduce(xb, yb)
    assert s1.dtype == torch.float32
    np.testing.assert_allclose(s1.cpu().numpy(), xr.sum(dim=(1, 2)).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(s2.cpu().numpy(), (xr * yr).sum(dim=(1, 2)).numpy(), rtol=1e-5, atol=1e-4)
    a1, a2, b = rng.normal(size=(3, 40)), rng.normal(size=(3, 40)), rng.normal(size=(3, 40))
    out = ops.nc_lin2(tuple(x.shape), xb, dev(a1), yb, dev(a2), dev(b), flags=1, slope=0.3)
    ref = t64(a1)[:, None, None, :] * O.leaky_relu(xr, 0.3) + t64(a2)[:, None, None, :] * yr + t64(b)[:, None, None, :]
    assert out.dtype == torch.bfloat16
    assert float((out.double().cpu() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())
    for fn, rf in ((lambda: ops.act_fwd(xb, 1, 0.2), O.leaky_relu(xr, 0.2)), (lambda: ops.axpby(xb, yb, 0.5, -2.0), 0.5 * xr - 2 * yr),
                   (lambda: ops.mul(xb, yb), xr * yr), (lambda: ops.act_bwd(yb, xb, 2), yr * (xr > 0))):
        got = fn()
        assert got.dtype == torch.bfloat16
        assert float((got.double().cpu() - rf).abs().max()) <= 2.0 ** -8 * float(rf.abs().max())
    got = ops.sqdiff_sum(xb, yb, 0.25)
    np.testing.assert_allclose(float(got), 0.25 * float(((xr - yr) ** 2).sum()), rtol=1e-5)
    # max pooling picks values, no arithmetic: exact; the backward routes each gradient to the first maximum
    for k, s, pad in ((2, 2, 0), (3, 2, 1)):
        xp = bf16_round(rng.normal(size=(2, 12, 10, 16))).requires_grad_(True)
        ref = O.maxpool(xp, k, s, pad)
        got = ops.maxpool_fwd(xp.detach().to(torch.bfloat16).cuda(), k, s, pad)
        assert torch.equal(got.double().cpu(), ref.detach())
        gy = bf16_round(rng.normal(size=tuple(ref.shape)))
        (ref * gy).sum().backward()
        gx = ops.maxpool_bwd(xp.detach().to(torch.bfloat16).cuda(), gy.to(torch.bfloat16).cuda(), k, s, pad)
        assert float((gx.double().cpu() - xp.grad).abs().max()) <= 2.0 ** -8 * float(xp.grad.abs().max())
    # round trip of the conversion kernel
    f = dev(x)
    assert torch.equal(ops.cast(f, torch.bfloat16), f.to(torch.bfloat16)) and torch.equal(ops.cast(xb, torch.float32), xb.float())


def test_bf16_generator_and_discriminator_loss_against_the_fp32_oracle():
    from confignet_amd.dnn_models.hologan_discriminator import HologanDiscriminator
    from confignet_amd.dnn_models.hologan_generator import HologanGenerator
    from confignet_amd.losses import compute_discriminator_loss
    rng = np.random.default_rng(128)
    g = HologanGenerator(43, (128, 128), 128, 2, "tanh", rng=rng)
    ws = g.get_weights()
    ws[1] = (1 + 0.5 * rng.standard_normal(32768)).astype(np.float32)
    g.set_weights(ws)
    z = rng.normal(size=(2, 43))
    rot = rng.uniform(-0.4, 0.4, size=(2, 3)).astype(np.float32)
    img = g((z, rot))
    assert img.dtype == torch.float32                                   # 3-channel images stay fp32
    wr = [t64(w) for w in g.get_weights()]
    ref = R.generator_forward(wr, t64(z), t64(rot), 128)
    # 15 bf16-stored layers: the error grows by ~0.1-0.3 % relative L2 per layer (scripts/dev/bf16_diag.py prints it layer by
    # layer: 0.28 % after the first Conv3D, 1.4 % at the image); single pixels of the tanh output deviate by up to ~0.1
    diff = img.detach().cpu().double() - ref
    rel, err = float(diff.norm() / ref.norm()), float(diff.abs().max())
    assert rel <= 3e-2 and err <= 0.2, "bf16 generator image: rel-L2 %.3e, max abs err %.3e (tanh output in [-1, 1])" % (rel, err)
    # The same comparison with the oracle ROUNDING WHAT THE PRODUCT ROUNDS (oracle.ref_ops.bf16_storage: every activation tensor
    # of more than 4 channels at the product's kernel boundaries and the filters of the bf16 convolutions to bf16, arithmetic in
    # float64): what is left is fp32 accumulation, the rounding of the pre-summed class filters of the upsample-folded layers
    # and values on a rounding boundary -- an order of magnitude below the storage error itself, so a kernel that is wrong by a
    # per cent in one layer shows here while the bound above would pass it
    # (the upsample-folded layers round PRE-SUMMED class filters, which the oracle does not restate: this comparison runs the
    # product with the collapse off -- the same bf16 kernels on per-tap filters -- and the collapsed product is held to it below)
    from confignet_amd import ops
    with O.bf16_storage():
        ref_b = R.generator_forward(wr, t64(z), t64(rot), 128).detach()
    upfold, ops.UPFOLD = ops.UPFOLD, False
    try:
        with torch.no_grad():
            img_d = g((z, rot)).detach().cpu().double()
    finally:
        ops.UPFOLD = upfold
    diff_b = img_d - ref_b
    rel_b, err_b = float(diff_b.norm() / ref_b.norm()), float(diff_b.abs().max())
    rel_c = float((img.detach().cpu().double() - img_d).norm() / img_d.norm())
    print("bf16 generator image vs the fp32 oracle: rel-L2 %.3e max %.3e; per-tap filters vs the bf16-storage oracle: rel-L2 %.3e max %.3e; "
          "class filters vs per-tap filters: rel-L2 %.3e" % (rel, err, rel_b, err_b, rel_c))
    assert rel_b <= BF16_EMU_GEN_REL and rel_c <= 3e-2, (rel_b, rel_c, rel)
    g.zero_grad()
    cot = rng.normal(size=tuple(ref.shape))
    torch.autograd.backward((img * dev(cot)).sum(), inputs=g.trainable_weights)
    wr = [w.requires_grad_(True) for w in wr]
    grads = torch.autograd.grad((R.generator_forward(wr, t64(z), t64(rot), 128) * t64(cot)).sum(), wr, allow_unused=True)
    for i, (p, gr) in enumerate(zip(g.weights, grads)):
        if gr is None or float(gr.norm()) == 0.0:
            continue
        assert p.grad.dtype == torch.float32
        # a pre-activation within the accumulated bf16 error of zero (~0.5 % of the elements per layer) takes the other LeakyReLU
        # branch than in the oracle; each such flip changes a gradient entry by the factor 0.3 <-> 1, which adds ~5 % relative L2
        # per activation layer in quadrature (14 % measured at the learned input, the deepest tensor).  The kernels themselves
        # are pinned at 2^-8 by the raw-op test above.
        got = p.grad.detach().cpu().double()
        rel = float((got - gr).norm() / gr.norm())
        cos = float((got * gr).sum() / (got.norm() * gr.norm()))
        assert rel <= 0.3 and cos >= 0.95, "bf16 generator grad[%d] %s: rel-L2 %.3e cos %.4f" % (i, tuple(p.shape), rel, cos)

    d = HologanDiscriminator((64, 64), 5, 512, 3, 48, True, rng=rng)
    real, fake = rng.uniform(-1, 1, size=(3, 64, 64, 3)), rng.uniform(-1, 1, size=(3, 64, 64, 3))
    # every head of the discriminator forward against the bf16-storage oracle (and, for scale, against the fp32 oracle)
    with torch.no_grad():
        outs = d(d.to_device(real))
    dw0 = [t64(w) for w in d.get_weights()]
    ref_f = R.discriminator_forward(dw0, t64(real))
    with O.bf16_storage():
        ref_e = R.discriminator_forward(dw0, t64(real))
    worst_f = worst_e = 0.0
    for k in ref_f:
        got = outs[k].detach().cpu().double().reshape(ref_f[k].shape)
        scale = max(1.0, float(ref_f[k].abs().max()))
        worst_f = max(worst_f, float((got - ref_f[k]).abs().max()) / scale)
        worst_e = max(worst_e, float((got - ref_e[k]).abs().max()) / scale)
    print("bf16 discriminator heads: max err vs the fp32 oracle %.3e, vs the bf16-storage oracle %.3e" % (worst_f, worst_e))
    assert worst_e <= BF16_EMU_DISCR_ERR, (worst_e, worst_f)
    d.zero_grad()
    losses = compute_discriminator_loss(d, d.to_device(real), d.to_device(fake))
    torch.autograd.backward(losses["loss_sum"], inputs=d.trainable_weights)
    dw = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in d.get_weights()]
    ref_l = S.discriminator_loss(dw, t64(real), t64(fake))
    for k in losses:
        v = float(ref_l[k].detach())
        assert abs(float(losses[k]) - v) <= 3e-2 * max(1.0, abs(v)), (k, float(losses[k]), v)
    gr = S.grads_of(ref_l["loss_sum"], dw)
    # (R1 differentiates an input gradient: two passes through the 5 blocks.  Tensors of a few elements -- the from-RGB bias is a
    # sum over every pixel of a cancelling quantity -- are compared only as part of the whole gradient vector.)
    for i, (p, r) in enumerate(zip(d.weights, gr)):
        got = p.grad.detach().cpu().double()
        if r.numel() < 1000:
            continue
        rel = float((got - r).norm() / (r.norm() + 1e-30))
        cos = float((got * r).sum() / (got.norm() * r.norm() + 1e-30))
        assert rel <= 0.3 and cos >= 0.95, "bf16 discriminator (R1) grad[%d] %s: rel-L2 %.3e cos %.4f" % (i, tuple(p.shape), rel, cos)
    got = torch.cat([p.grad.detach().cpu().double().reshape(-1) for p in d.weights])
    ref = torch.cat([r.reshape(-1) for r in gr])
    # measured: rel-L2 0.24, cos 0.972 (run-to-run +-0.002: atomics order); the branch flips above, through two passes
    assert float((got - ref).norm() / ref.norm()) <= 0.3 and float((got * ref).sum() / (got.norm() * ref.norm())) >= 0.96


def test_bf16_second_stage_iteration_runs_under_graph_dispatch_and_tracks_the_fp32_run():
    """One model, the same batches: the fp32 iteration and the bf16 iteration (eager, then three graph iterations) give the
    same loss scalars to 5 % (random-init networks amplify rounding through ~60 layers; the sharp bf16 checks are the raw-op
    and single-network tests above)."""
    from confignet_amd import ConfigNet, SyntheticFaceDataset, ops, optim
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.confignet_utils import merge_configs
    ds = SyntheticFaceDataset(16, 128, seed=3)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "output_shape": (128, 128, 3)})
    ds.process_metadata(cfg, True)
    out = {}
    for mode in ("f32", "bf16"):
        ops.set_activation_dtype(mode)
        np.random.seed(5)
        m = ConfigNet(cfg, seed=0)
        m.setup_training(None, ds, 0, real_training_set=ds)
        dopt, gopt = optim.Adam(**m.config["optimizer"]), optim.Adam(**m.config["optimizer"])
        out[mode] = [{k: float(v) for k, v in d.items()} for d in m.training_iteration(ds, ds, dopt, gopt)]
        if mode == "bf16":
            m.use_graphs = True
            for _ in range(3):
                last = m.training_iteration(ds, ds, dopt, gopt)
            assert all(np.isfinite(float(d["loss_sum"])) for d in last)
            assert all(g.graph is not None for g in m._graphs.values())
            assert m.generator.weights[2].dtype == torch.float32 and m.generator.grad_arena.dtype == torch.float32   # fp32 master
    # A discriminator head on FAKE images sees the bf16 generator's image error (rel-L2 <= 3e-2, test above) multiplied by the
    # head's input-gradient norm -- which the R1 term measures: gp_loss_i = 5 mean |d D_i / d x|^2 (246 for the deepest head
    # of a random-init discriminator, i.e. |grad| ~ 7 per unit image change).  The allowance for those keys scales with it.
    for a, b in zip(out["f32"], out["bf16"]):
        assert a.keys() == b.keys()
        extra = {k: 0.02 * np.sqrt(a["gp_loss_" + k.rsplit("_", 1)[1]]) for k in a if k.startswith("GAN_loss_fake_") and "gp_loss_0" in a}
        if "gp_loss_0" not in a:       # generator step: the same heads on generated images, without an R1 term to scale by
            extra = {k: 0.1 * max(1.0, abs(a[k])) for k in a if k.startswith("GAN_loss_")}
        extra["loss_sum"] = sum(extra.values())
        for k in a:
            assert abs(a[k] - b[k]) <= 5e-2 * max(1.0, abs(a[k])) + extra.get(k, 0.0), (k, a[k], b[k])


def test_bf16_data_parallel_dispatch_at_full_size_matches_cpu_oracle(tmp_path):
    """BASELINE.json configs[2]'s per-rank workload as ONE test (round 3 tested bf16 and the data-parallel dispatch apart): the
    second-stage iteration at 256x256, batch 16, bf16 compute, step graphs + cross-iteration overlap, on a 1-rank RCCL group
    (CN_FORCE_DP=1) -- with and without config["dp_global_batch_statistics"] -- and in the single-process dispatch:
    * every one of the 60 loss scalars of a whole iteration against the fp32 CPU oracle on the same weights and batches, at the
      bf16 bound of this file (5e-2 of the scalar; the heads on fake images by the R1-scaled allowance);
    * each of the three runs on its own weights and batches (runs are not compared with each other: see the end of the test)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    helper = os.path.join(root, "tests", "dp_bf16_helper.py")
    dp_env = {"CN_FORCE_DP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29549", "RANK": "0", "WORLD_SIZE": "1"}
    runs = {}
    for tag, extra, flag in (("single", {}, 0), ("dp", dp_env, 0), ("dp_stats", dp_env, 1)):
        env = {k: v for k, v in os.environ.items() if k not in ("CN_FORCE_DP",)}
        env.update(extra)
        path = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, helper, path, str(flag), "bf16"], env=env, cwd=root, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        runs[tag] = (json.load(open(path + ".json")), np.load(path))
    assert not runs["single"][0]["dp"] and runs["dp"][0]["dp"] and all(runs["dp"][0]["split"]) and not any(runs["single"][0]["split"])
    for tag, (info, _) in runs.items():
        assert "pre-replayed" in info["dispatch"], (tag, info["dispatch"])
        n = 0
        for got, step in zip(info["losses"], ("d", "synth_d", "latent_d", "g")):
            ref = info["ref"][step]
            assert list(got.keys()) == list(ref.keys())
            extra = {k: 0.02 * np.sqrt(ref["gp_loss_" + k.rsplit("_", 1)[1]]) for k in ref if k.startswith("GAN_loss_fake_") and "gp_loss_0" in ref}
            if "gp_loss_0" not in ref:
                extra = {k: 0.1 * max(1.0, abs(ref[k])) for k in ref if k.startswith("GAN_loss_")}
            # (5 x the latent discriminator's loss on bf16-encoded latents, ~15 after five training iterations: 5.5 % seen once in
            # six runs of one build, 2 - 4 % otherwise)
            if "latent_GAN_loss" in ref:
                extra["latent_GAN_loss"] = 3e-2 * abs(ref["latent_GAN_loss"])
            extra["loss_sum"] = sum(extra.values())
            for k in ref:
                n += 1
                assert np.isfinite(got[k]) and abs(got[k] - ref[k]) <= 5e-2 * max(1.0, abs(ref[k])) + extra.get(k, 0.0), (tag, step, k, got[k], ref[k])
        assert n == 19 + 19 + 4 + len(info["losses"][3])
    # (No run-against-run comparison: each run trains five iterations before the compared one, the bf16 kernels' atomics are not
    # covered by the deterministic mode, and lr * sign(g) steps of a fresh GAN make two runs of ONE configuration drift apart by more
    # than any useful bound (0.33 against 0.23 on a real-image head measured).  Every run is held to the oracle ON ITS OWN weights
    # and batches above; the bit-for-bit statement about the dispatch itself is
    # test_steps_gpu.py::test_data_parallel_dispatch_with_overlap_matches_single_process in fp32 deterministic mode.)
