"""Pins for the oracle: (1) torch restatement vs independent NumPy-loop restatement,
(2) analytic known-answer tests that need no TensorFlow."""
import math

import numpy as np
import pytest
import torch

from oracle import np_ops as NP
from oracle import ref_nets as R
from oracle import ref_ops as O

torch.manual_seed(0)


def t(a, dt=torch.float64):
    return torch.tensor(np.asarray(a), dtype=dt)


@pytest.mark.parametrize("shape,k,s", [((2, 7, 6, 3), (4, 4), 1), ((2, 8, 8, 5), (3, 3), 2), ((1, 9, 7, 2), (3, 3), 2),
                                       ((2, 5, 4, 6, 3), (3, 3, 3), 1), ((1, 6, 6, 4), (1, 1), 1), ((1, 11, 11, 3), (7, 7), 2)])
def test_conv_same_double_implementation(shape, k, s):
    rng = np.random.default_rng(1)
    x = rng.normal(size=shape)
    w = rng.normal(size=(*k, shape[-1], 5))
    b = rng.normal(size=5)
    y_t = O.conv_same(t(x), t(w), t(b), stride=s).numpy()
    y_n = NP.conv_same(x, w, b, stride=s)
    assert y_t.shape == y_n.shape
    np.testing.assert_allclose(y_t, y_n, atol=1e-10)
    # SAME output size = ceil(in/stride)
    assert y_t.shape[1:-1] == tuple(math.ceil(d / s) for d in shape[1:-1])


def test_same_pad_rules():
    assert O.same_pad(16, 4, 1)[:2] == (1, 2)      # k4 s1 -> (1,2)
    assert O.same_pad(256, 3, 2)[:2] == (0, 1)     # k3 s2 even input -> (0,1)
    assert O.same_pad(32, 3, 1)[:2] == (1, 1)


def test_norms_double_implementation():
    rng = np.random.default_rng(2)
    x = rng.normal(size=(2, 5, 6, 4)) * 3 + 1
    g, b = rng.normal(size=4), rng.normal(size=4)
    np.testing.assert_allclose(O.instance_norm(t(x), t(g), t(b)).numpy(), NP.instance_norm(x, g, b), atol=1e-12)
    mu, sd = O.layer_style(t(x))
    mu_n, sd_n = NP.layer_style(x)
    np.testing.assert_allclose(mu.numpy(), mu_n, atol=1e-12)
    np.testing.assert_allclose(sd.numpy(), sd_n, atol=1e-12)


def test_adain_statistics():
    """AdaIN output has per-(n,c) mean b and variance (s+1)^2 * var/(var+1e-3)."""
    rng = np.random.default_rng(3)
    x = t(rng.normal(size=(2, 4, 4, 4, 6)) * 2 + 0.5)
    z = t(rng.normal(size=(2, 7)))
    mlp = [t(rng.normal(size=s)) for s in [(7, 5), (5,), (5, 12), (12,)]]
    y = O.adain(x, z, mlp)
    sb = O.mlp_simple(z, mlp, 0.2).reshape(2, 2, 6)
    var = x.var(dim=(1, 2, 3), unbiased=False)
    np.testing.assert_allclose(y.mean(dim=(1, 2, 3)).numpy(), sb[:, 1].numpy(), atol=1e-10)
    np.testing.assert_allclose(y.var(dim=(1, 2, 3), unbiased=False).numpy(),
                               ((sb[:, 0] + 1) ** 2 * var / (var + 1e-3)).numpy(), atol=1e-10)
    # vs NumPy layer norm
    yn = NP.layer_norm_spatial(x.numpy()) * (sb[:, 0].numpy()[:, None, None, None] + 1) + sb[:, 1].numpy()[:, None, None, None]
    np.testing.assert_allclose(y.numpy(), yn, atol=1e-10)


def test_euler_and_rotation_kat():
    assert np.allclose(O.euler_angles_to_matrix(torch.zeros(2, 3, dtype=torch.float64)).numpy(), np.eye(3))
    a = t(np.random.default_rng(4).uniform(-0.5, 0.5, size=(3, 3)))
    Rm = O.euler_angles_to_matrix(a)
    np.testing.assert_allclose((Rm @ Rm.transpose(1, 2)).numpy(), np.tile(np.eye(3), (3, 1, 1)), atol=1e-12)
    np.testing.assert_allclose(Rm.numpy(), NP.euler_angles_to_matrix(a.numpy()), atol=1e-12)
    # identity rotation => exact identity resample
    g = t(np.random.default_rng(5).normal(size=(2, 6, 6, 6, 3)))
    np.testing.assert_array_equal(O.transform_3d_grid(g, torch.eye(3, dtype=torch.float64).expand(2, 3, 3)).numpy(), g.numpy())
    # general rotation vs per-voxel loop
    out = O.transform_3d_grid(g, Rm[:2])
    np.testing.assert_allclose(out.numpy(), NP.transform_3d_grid(g.numpy(), Rm[:2].numpy()), atol=1e-12)


def test_rotation_90deg_is_permutation_with_clamp():
    """90 degrees about axis 0 maps voxel (i,j,k) <- source (i, c+(k-c)... ) exactly: a one-hot
    voxel moves to the permuted index (interior => no clamping involved)."""
    g = np.zeros((1, 4, 4, 4, 1))
    g[0, 1, 2, 0, 0] = 1.0
    Rm = NP.euler_angles_to_matrix([[math.pi / 2, 0, 0]])
    out = O.transform_3d_grid(t(g), t(Rm)).numpy()
    ref = NP.transform_3d_grid(g, Rm)
    np.testing.assert_allclose(out, ref, atol=1e-12)
    assert abs(out.sum() - 1.0) < 1e-9 and (out > 0.5).sum() == 1


def test_gan_losses_and_r1_kat():
    z = torch.zeros(5, 1, dtype=torch.float64)
    assert abs(O.gan_g_loss(z).item() - math.log(2)) < 1e-12
    assert abs(O.gan_d_loss(torch.ones_like(z), z).item() - math.log(2)) < 1e-12
    # R1 of a linear discriminator out = x.w  is 5*||w||^2
    w = t(np.random.default_rng(6).normal(size=(12,)))
    x = t(np.random.default_rng(7).normal(size=(4, 12))).requires_grad_(True)
    out = (x @ w).reshape(4, 1)
    assert abs(O.r1_penalty(out, x).item() - 5 * (w ** 2).sum().item()) < 1e-10


def test_keras_adam_first_step_and_shared_counter():
    """First Keras-Adam step = lr*sign(g) when |g| >> eps (confirmed by the reference's finetune
    fixture: rotation deltas of +-1.000e-4 at lr 1e-4); the step counter is per optimizer."""
    p1 = torch.tensor([1.0, -2.0, 3.0], dtype=torch.float64)
    p2 = torch.tensor([0.5], dtype=torch.float64)
    opt = O.KerasAdam(lr=4e-4, beta_1=0.0, beta_2=0.9)
    g1 = torch.tensor([0.3, -7.0, 1e-2], dtype=torch.float64)
    before = p1.clone()
    opt.apply_gradients([(g1, p1)])
    np.testing.assert_allclose((before - p1).numpy(), 4e-4 * np.sign(g1.numpy()), rtol=1e-4)
    # second apply on ANOTHER variable uses t=2: lr_t = lr*sqrt(1-0.9^2), v=(1-b2) g^2 -> step = lr_t/sqrt(0.1)
    g2 = torch.tensor([2.0], dtype=torch.float64)
    b2 = p2.clone()
    opt.apply_gradients([(g2, p2)])
    expect = 4e-4 * math.sqrt(1 - 0.9 ** 2) * 2.0 / (math.sqrt(0.1 * 4.0) + 1e-7)
    assert abs((b2 - p2).item() - expect) < 1e-12


def _rand_weights(shapes, seed, dt=torch.float64):
    rng = np.random.default_rng(seed)
    return [torch.tensor(O.glorot_uniform(rng, s) if len(s) > 1 else rng.normal(size=s) * 0.1, dtype=dt) for s in shapes]


def test_learned_input_is_ones_and_kernel_grad_zero():
    shapes = R.generator_weight_shapes(9, 128, n_mlp_units=8)
    w = _rand_weights(shapes, 8)
    w[0] = torch.zeros(1, 32768, dtype=torch.float64, requires_grad=True)
    w[1] = torch.ones(32768, dtype=torch.float64, requires_grad=True)
    z = t(np.random.default_rng(9).normal(size=(1, 9)))
    img = R.generator_forward(w, z, torch.zeros(1, 3, dtype=torch.float64), 128)
    assert img.shape == (1, 128, 128, 3) and img.abs().max() <= 1.0
    gk, gb = torch.autograd.grad(img.sum(), [w[0], w[1]])
    assert gk.abs().max().item() == 0.0 and gb.abs().max().item() > 0


def test_discriminator_dict_order_and_shapes():
    w = _rand_weights(R.discriminator_weight_shapes(64), 10)
    out = R.discriminator_forward(w, t(np.random.default_rng(11).normal(size=(2, 64, 64, 3))))
    assert list(out.keys()) == ["discr_style_%d" % i for i in range(5)] + ["discr_final"]
    assert all(v.shape == (2, 1) for v in out.values())
    lr = R.latent_regressor_forward(_rand_weights(R.latent_regressor_weight_shapes(9, 64), 12),
                                    t(np.random.default_rng(13).normal(size=(2, 64, 64, 3))))
    assert lr.shape == (2, 12)


def test_reference_golden_weight_free_facts():
    """What the reference's own goldens tell us without weights: the
    fine-tune fixture moves exactly the blendshape slice 7..36 of the 144-d latent."""
    import os
    p = "/root/reference/tests/test_assets"
    if not os.path.isdir(p):
        pytest.skip("reference assets not present (GPU box)")
    base = np.load(os.path.join(p, "confignet_basic_ref_256.npz"))
    ft = np.load(os.path.join(p, "confignet_finetune_ref_256.npz"))
    keys_b, keys_f = list(base.keys()), list(ft.keys())
    emb_b = base["embedding"]
    emb_f = ft[[k for k in keys_f if "embedding" in k][0]]
    assert emb_b.shape == (1, 144)
    # the other 114 entries agree to float noise (~1e-7: two separate encoder runs)
    moved = np.nonzero(np.abs(emb_b[0] - emb_f[0]) > 1e-5)[0]
    assert moved.min() == 7 and moved.max() == 36 and len(moved) == 30
    np.testing.assert_allclose(np.abs(emb_b[0, moved] - emb_f[0, moved]), 1e-4, rtol=2e-2)


def test_oracle_matches_tensorflow_pins():
    """Consumes tests/golden/tf_pins.npz -- written OUT OF LOOP by scripts/tf_pin_dump.py on a machine that has TensorFlow
    2.x and the reference checkout -- and checks every [TF-2.1] rule the oracle hard-codes against executed TensorFlow:
    Conv2dAdaIn / Conv3dAdaIn / DiscrBlock outputs (SAME padding split, LeakyReLU 0.3 / 0.2, LayerNormalization eps 1e-3,
    eps-on-std instance norm, style eps), the 3-D resampling incl. both gradients, the shared-counter Keras-Adam trace and
    the keras.applications weight orders.  Skipped while the file does not exist (TensorFlow cannot be installed here):
    until someone commits it, parity is against this repository's own restatement."""
    import os
    import pytest
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_pins.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/tf_pins.npz not present: run scripts/tf_pin_dump.py where TensorFlow is available")
    _check_pins(np.load(path, allow_pickle=False))


def _check_pins(P):
    files = P.files if hasattr(P, "files") else list(P.keys())
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    unpack = lambda prefix: [t(P["%s_%d" % (prefix, i)]) for i in range(int(P[prefix + "_n"]))]
    for name in ("conv2d_adain", "conv3d_adain"):
        ck, cb, *mlp = unpack(name + "_w")
        y = O.adain(O.leaky_relu(O.conv_same(t(P[name + "_x"]), ck, cb), 0.3), t(P[name + "_z"]), mlp, 0.2)
        assert float((y - t(P[name + "_y"])).abs().max()) < 1e-4, name
    for tag in ("even", "odd"):
        w = unpack("discr_block_%s_w" % tag)                               # conv kernel, bias, gamma, beta
        y, st = R.discr_block(t(P["discr_block_%s_x" % tag]), *w, return_styles=True)
        assert float((y - t(P["discr_block_%s_y" % tag])).abs().max()) < 1e-4 and float((st - t(P["discr_block_%s_style" % tag])).abs().max()) < 1e-4
    grid, ang = t(P["rot_grid"]).requires_grad_(True), t(P["rot_angles"]).requires_grad_(True)
    Rm = O.euler_angles_to_matrix(ang)
    out = O.transform_3d_grid(grid, Rm)
    g_grid, g_ang = torch.autograd.grad((out * t(P["rot_cot"])).sum(), [grid, ang])
    assert float((Rm - t(P["rot_matrix"])).abs().max()) < 1e-6 and float((out - t(P["rot_out"])).abs().max()) < 1e-4
    assert float((g_grid - t(P["rot_g_grid"])).abs().max()) < 1e-4
    assert float((g_ang - t(P["rot_g_angles"])).abs().max()) < 1e-3 * max(1.0, float(t(P["rot_g_angles"]).abs().max()))
    gamma, beta = unpack("inorm_w")
    assert float((O.instance_norm(t(P["inorm_x"]), gamma, beta) - t(P["inorm_y"])).abs().max()) < 1e-5
    mu, sd = O.layer_style(t(P["inorm_x"]))
    assert float((mu - t(P["style_mean"])).abs().max()) < 1e-6 and float((sd - t(P["style_std"])).abs().max()) < 1e-6
    # one optimizer, three variables in turn, two rounds: t = 1..6 (R10)
    opt = O.KerasAdam(lr=4e-4, beta_1=0.0, beta_2=0.9)
    vs = [t(v).clone() for v in P["adam_theta0"]]
    k = 0
    for it in range(2):
        for j, v in enumerate(vs):
            opt.apply_gradients([(t(P["adam_grads"][3 * it + j]), v)])
            assert float((v - t(P["adam_trace"][k])).abs().max()) < 1e-7, (it, j)
            k += 1
    assert int(P["adam_iterations"]) == 6 == opt.t
    opt2, v = O.KerasAdam(lr=1e-4), t(P["adam_theta0"][0]).clone()
    for it in range(3):
        opt2.apply_gradients([(t(P["adam_grads"][it]), v)])
        assert float((v - t(P["adam_default_trace"][it])).abs().max()) < 1e-7
    for k_, s_, n_ in ((4, 1, 8), (3, 2, 8), (3, 2, 9), (3, 1, 8)):
        x = torch.zeros(1, n_, n_, 1, dtype=torch.float64)
        x[0, 0, 0, 0], x[0, n_ - 1, n_ - 1, 0] = 1.0, 2.0
        y = O.conv_same(x, torch.ones(k_, k_, 1, 1, dtype=torch.float64), None, stride=s_)
        assert torch.equal(y, t(P["same_k%d_s%d_n%d" % (k_, s_, n_)])), (k_, s_, n_)
    assert abs(float(P["keras_leakyrelu_of_minus1"][0]) + 0.3) < 1e-7 and abs(float(P["tf_nn_leaky_relu_of_minus1"][0]) + 0.2) < 1e-7
    assert abs(float(P["layernorm_default_eps"]) - 1e-3) < 1e-12
    if "generator_weight_shapes" in files:                    # whole networks: get_weights() order and a seeded forward (R11)
        assert [eval(str(s_)) for s_ in P["generator_weight_shapes"]] == [tuple(s_) for s_ in R.generator_weight_shapes(9, 128, n_mlp_units=8)], \
            "HologanGenerator.get_weights() order differs from oracle/ref_nets.py:generator_weight_shapes"
        img = R.generator_forward(unpack("generator_w"), t(P["generator_z"]), t(P["generator_rot"]), 128)
        assert float((img - t(P["generator_img"])).abs().max()) < 1e-3, "generator forward vs TensorFlow"
        assert [eval(str(s_)) for s_ in P["discriminator_weight_shapes"]] == [tuple(s_) for s_ in R.discriminator_weight_shapes(64)]
        o = R.discriminator_forward(unpack("discriminator_w"), t(P["discriminator_x"]))
        assert [str(k_) for k_ in P["discriminator_out_keys"]] == list(o.keys())
        got = torch.cat([v.reshape(2, 1) for v in o.values()], dim=1)
        assert float((got - t(P["discriminator_out"])).abs().max()) < 1e-3 * max(1.0, float(t(P["discriminator_out"]).abs().max()))
    if "resnet50_weight_names" in files:
        names = [str(n) for n in P["resnet50_weight_names"]]
        want = []
        for lname, kind, *_ in R.resnet50_layer_order():
            want += [lname + "/kernel:0", lname + "/bias:0"] if kind == "conv" else \
                [lname + "/gamma:0", lname + "/beta:0", lname + "/moving_mean:0", lname + "/moving_variance:0"]
        assert names == want, "keras ResNet50 get_weights() order differs from oracle/ref_nets.py:resnet50_layer_order"
        assert [eval(s) for s in P["resnet50_weight_shapes"]] == [tuple(s) for s in R.resnet50_weight_shapes()]
        assert abs(float(P["resnet50_bn_eps"][0]) - R.BN_EPS) < 1e-12
        assert np.allclose(P["caffe_preprocess_probe"][0, 0, 0], [30.0 - 103.939, 20.0 - 116.779, 10.0 - 123.68])
    if "inception_v3_weight_shapes" in files:
        from oracle import ref_metrics as M
        assert [eval(s) for s in P["inception_v3_weight_shapes"]] == M.inception_weight_shapes(), \
            "keras InceptionV3 get_weights() order differs from oracle/ref_metrics.py"
        if "inception_v3_probe_features" in files:             # the seeded forward: same RandomState draws as the dump script
            rs = np.random.RandomState(int(P["inception_v3_probe_seed"]))
            names = [str(n) for n in P["inception_v3_weight_names"]]
            ws = [(rs.uniform(0.5, 1.5, size=s_) if "moving_variance" in n else rs.normal(size=s_) * (0.05 if len(s_) == 4 else 0.1)).astype(np.float32)
                  for s_, n in zip(M.inception_weight_shapes(), names)]
            xin = rs.uniform(-1, 1, size=(2, 139, 107, 3)).astype(np.float32)
            assert np.array_equal(xin, P["inception_v3_probe_input"])
            got = M.inception_features([t(w) for w in ws], t(xin)).numpy()
            assert np.abs(got - P["inception_v3_probe_features"]).max() <= 1e-3 * np.abs(got).max()
            assert np.allclose(P["inception_preprocess_probe"], [[-1.0, 0.0, 1.0]])


def test_pin_checker_runs_on_a_self_made_file():
    """The checker above is exercised here on a pins dictionary produced BY THE ORACLE in the layout scripts/tf_pin_dump.py
    writes (so a typo in a key or a shape convention shows up now, not on the day a TensorFlow-made file arrives).  This
    pins nothing: it only proves the consumer works."""
    rng = np.random.default_rng(0)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    P = {}

    def pack(prefix, arrays):
        P[prefix + "_n"] = np.array(len(arrays))
        for i, a in enumerate(arrays):
            P["%s_%d" % (prefix, i)] = np.asarray(a)
    for name, xs, k in (("conv2d_adain", (2, 9, 8, 6), (4, 4)), ("conv3d_adain", (2, 5, 4, 6, 3), (3, 3, 3))):
        x, z = rng.standard_normal(xs), rng.standard_normal((2, 7))
        ws = [rng.standard_normal(sh) * 0.3 for sh in ((*k, xs[-1], 8), (8,), (7, 5), (5,), (5, 16), (16,))]
        P[name + "_x"], P[name + "_z"] = x, z
        P[name + "_y"] = O.adain(O.leaky_relu(O.conv_same(t(x), t(ws[0]), t(ws[1])), 0.3), t(z), [t(w) for w in ws[2:]], 0.2).numpy()
        pack(name + "_w", ws)
    for tag, hw in (("even", (8, 10)), ("odd", (9, 7))):
        x = rng.standard_normal((2, *hw, 5))
        ws = [rng.standard_normal(sh) * 0.3 for sh in ((3, 3, 5, 6), (6,), (6,), (6,))]
        y, st = R.discr_block(t(x), *[t(w) for w in ws], return_styles=True)
        P["discr_block_%s_x" % tag], P["discr_block_%s_y" % tag], P["discr_block_%s_style" % tag] = x, y.numpy(), st.numpy()
        pack("discr_block_%s_w" % tag, ws)
    grid = t(rng.standard_normal((2, 16, 16, 16, 3))).requires_grad_(True)
    ang = t([[0.3, -0.1, 0.05], [-0.45, 0.15, 0.0]]).requires_grad_(True)
    cot = rng.standard_normal((2, 16, 16, 16, 3))
    Rm = O.euler_angles_to_matrix(ang)
    out = O.transform_3d_grid(grid, Rm)
    gg, ga = torch.autograd.grad((out * t(cot)).sum(), [grid, ang])
    P.update(rot_grid=grid.detach().numpy(), rot_angles=ang.detach().numpy(), rot_matrix=Rm.detach().numpy(), rot_out=out.detach().numpy(),
             rot_cot=cot, rot_g_grid=gg.numpy(), rot_g_angles=ga.numpy())
    x = rng.standard_normal((2, 6, 5, 4))
    ws = [rng.standard_normal(4), rng.standard_normal(4)]
    P["inorm_x"], P["inorm_y"] = x, O.instance_norm(t(x), t(ws[0]), t(ws[1])).numpy()
    pack("inorm_w", ws)
    mu, sd = O.layer_style(t(x))
    P["style_mean"], P["style_std"] = mu.numpy(), sd.numpy()
    th, gs = rng.standard_normal((3, 50)), rng.standard_normal((6, 50))
    P["adam_theta0"], P["adam_grads"] = th, gs
    opt, vs, trace = O.KerasAdam(lr=4e-4, beta_1=0.0, beta_2=0.9), [t(v).clone() for v in th], []
    for it in range(2):
        for j, v in enumerate(vs):
            opt.apply_gradients([(t(gs[3 * it + j]), v)])
            trace.append(v.numpy().copy())
    P["adam_trace"], P["adam_iterations"] = np.stack(trace), np.array(6)
    opt2, v, t2 = O.KerasAdam(lr=1e-4), t(th[0]).clone(), []
    for it in range(3):
        opt2.apply_gradients([(t(gs[it]), v)])
        t2.append(v.numpy().copy())
    P["adam_default_trace"] = np.stack(t2)
    for k_, s_, n_ in ((4, 1, 8), (3, 2, 8), (3, 2, 9), (3, 1, 8)):
        x = torch.zeros(1, n_, n_, 1, dtype=torch.float64)
        x[0, 0, 0, 0], x[0, n_ - 1, n_ - 1, 0] = 1.0, 2.0
        P["same_k%d_s%d_n%d" % (k_, s_, n_)] = O.conv_same(x, torch.ones(k_, k_, 1, 1, dtype=torch.float64), None, stride=s_).numpy()
    P["keras_leakyrelu_of_minus1"], P["tf_nn_leaky_relu_of_minus1"], P["layernorm_default_eps"] = np.array([-0.3]), np.array([-0.2]), np.array(1e-3)
    gshapes = R.generator_weight_shapes(9, 128, n_mlp_units=8)
    gw = [rng.standard_normal(sh) * (0.3 if len(sh) > 1 else 0.1) for sh in gshapes]
    z, rot = rng.standard_normal((1, 9)), np.array([[0.2, -0.1, 0.0]])
    P["generator_weight_shapes"] = np.array([str(tuple(sh)) for sh in gshapes])
    P["generator_z"], P["generator_rot"] = z, rot
    P["generator_img"] = R.generator_forward([t(w) for w in gw], t(z), t(rot), 128).numpy()
    pack("generator_w", gw)
    dshapes = R.discriminator_weight_shapes(64)
    dw = [rng.standard_normal(sh) * (0.3 if len(sh) > 1 else 0.1) for sh in dshapes]
    x = rng.uniform(-1, 1, (2, 64, 64, 3))
    o = R.discriminator_forward([t(w) for w in dw], t(x))
    P["discriminator_weight_shapes"] = np.array([str(tuple(sh)) for sh in dshapes])
    P["discriminator_x"], P["discriminator_out_keys"] = x, np.array(list(o.keys()))
    P["discriminator_out"] = torch.cat([v.reshape(2, 1) for v in o.values()], dim=1).numpy()
    pack("discriminator_w", dw)
    _check_pins(P)


def test_forced_branch_decisions():
    """oracle.ref_ops.BranchControl(forced=...) (the instrument behind the whole-step gradient tests): a logged product mask /
    pool input that disagrees with the float64 pass only on near-zero inputs / near-tied windows is taken over; anything else is
    reported as unmatched if a gradient reaches it; without gradients nothing is looked up."""
    from oracle import ref_ops as O
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 8, 8, 4, dtype=torch.float64, generator=g)
    x.view(-1)[5] = 1e-9                                   # the GPU sees this one on the other side of zero
    x.view(-1)[6] = -1e-9
    x = x.requires_grad_(True)
    mask = (x.detach() > 0).reshape(-1).clone()
    mask[5], mask[6] = False, True
    O.BranchControl.start(forced=[("act", mask)])
    try:
        y = O.leaky_relu(x, 0.2)
        y.sum().backward()
        rep = O.BranchControl.forced_report()
    finally:
        O.BranchControl.stop()
    assert rep["forced_calls"] == 1 and rep["forced_decisions"] == 2 and not rep["unmatched"]
    assert float(x.grad.view(-1)[5]) == 0.2 and float(x.grad.view(-1)[6]) == 1.0
    assert torch.equal(x.grad.view(-1)[7:], torch.where(x.detach().view(-1)[7:] > 0, torch.tensor(1.0, dtype=torch.float64), torch.tensor(0.2, dtype=torch.float64)))
    # a mask that differs on a clearly non-zero input is NOT taken
    bad = (x.detach() > 0).reshape(-1).clone()
    big = int(x.detach().abs().reshape(-1).argmax())
    bad[big] = not bool(bad[big])
    x.grad = None
    O.BranchControl.start(forced=[("act", bad)])
    try:
        O.relu(x).sum().backward()
        rep = O.BranchControl.forced_report()
    finally:
        O.BranchControl.stop()
    assert rep["forced_calls"] == 0 and len(rep["unmatched"]) == 1
    # the product ran the two halves of this call's stack as separate launches (either order in the log): assembled per block
    x.grad = None
    good = (x.detach() > 0).reshape(2, -1).clone()
    good[0, 5], good[0, 6] = False, True
    O.BranchControl.start(forced=[("act", good[1].clone()), ("act", good[0].clone())])
    try:
        O.leaky_relu(x, 0.2).sum().backward()
        rep = O.BranchControl.forced_report()
    finally:
        O.BranchControl.stop()
    assert rep["forced_calls"] == 1 and rep["forced_decisions"] == 2 and not rep["unmatched"]
    assert float(x.grad.view(-1)[5]) == 0.2 and float(x.grad.view(-1)[6]) == 1.0
    # max-pool: a near-tie resolved the product's way (its fp32 input rounds the two candidates the other way round)
    p = torch.zeros(1, 2, 2, 1, dtype=torch.float64)
    p[0, 0, 0, 0], p[0, 0, 1, 0] = 1.0, 1.0 + 1e-12      # float64 winner: element 1; in fp32 both are 1.0 -> first maximum: element 0
    p = p.requires_grad_(True)
    O.BranchControl.start(forced=[("pool", p.detach().float(), (2, 2, 0))])
    try:
        O.maxpool(p, 2, 2).sum().backward()
        rep = O.BranchControl.forced_report()
    finally:
        O.BranchControl.stop()
    assert rep["forced_decisions"] == 1 and not rep["unmatched"]
    assert p.grad.reshape(-1).tolist() == [1.0, 0.0, 0.0, 0.0]


# ---- whole networks and DERIVATIVES against the independent NumPy restatement (oracle/np_nets.py) ----------------------------
def _np_weights(shapes, rng, scale=1.0):
    return [rng.normal(size=s) * (scale / np.sqrt(max(1, int(np.prod(s[:-1]))))) if len(s) > 1 else rng.normal(size=s) * 0.1 for s in shapes]


def test_generator_forward_against_numpy_restatement():
    """oracle.ref_nets.generator_forward (torch) == oracle.np_nets.generator_forward (tap loops, per-voxel resampling, written
    from hologan_generator.py on its own) on one seeded sample at the smallest resolution: max abs 1e-9 of a tanh image."""
    from oracle import np_nets as NN
    from oracle import ref_nets as R
    rng = np.random.default_rng(3)
    L, res = 12, 128
    ws = _np_weights(R.generator_weight_shapes(L, res), rng)
    ws[0] = np.zeros_like(ws[0])
    ws[1] = rng.normal(size=ws[1].shape)
    z, rot = rng.normal(size=(1, L)), np.array([[0.3, -0.2, 0.1]])
    ref = R.generator_forward([torch.tensor(w) for w in ws], torch.tensor(z), torch.tensor(rot), res).numpy()
    got = NN.generator_forward(ws, z, rot, res)
    assert got.shape == ref.shape == (1, res, res, 3)
    assert np.abs(ref).max() > 1e-3
    np.testing.assert_allclose(got, ref, atol=1e-9)


def test_discriminator_forward_against_numpy_restatement():
    from oracle import np_nets as NN
    from oracle import ref_nets as R
    rng = np.random.default_rng(4)
    res = 32
    ws = _np_weights(R.discriminator_weight_shapes(res), rng, 2.0)
    img = rng.uniform(-1, 1, size=(2, res, res, 3))
    ref = R.discriminator_forward([torch.tensor(w) for w in ws], torch.tensor(img))
    got = NN.discriminator_forward(ws, img)
    assert list(ref.keys()) == ["discr_style_%d" % i for i in range(5)] + ["discr_final"]
    for r, g in zip(ref.values(), got):
        np.testing.assert_allclose(g, r.numpy(), atol=1e-10)


def _block_case(seed):
    """A small DiscrBlock (3 -> 4 channels, 8 x 8 input) whose LeakyReLU inputs all keep a distance from zero: finite
    differences with steps far below that distance never cross a kink."""
    from oracle import np_ops as NP
    for s in range(seed, seed + 200):
        rng = np.random.default_rng(s)
        x = rng.uniform(-1, 1, size=(2, 8, 8, 3))
        ws = [rng.normal(size=(3, 3, 3, 4)) * 0.4, rng.normal(size=4) * 0.2, 1 + 0.3 * rng.normal(size=4), 0.2 * rng.normal(size=4)]
        head = rng.normal(size=(8, 1))
        if np.abs(NP.conv_same(x, ws[0], ws[1], stride=2)).min() > 2e-2:
            return x, ws, head, rng
    raise AssertionError("no kink-free case found")


def test_discr_block_backward_against_finite_differences_of_the_numpy_restatement():
    """First derivatives of DiscrBlock (conv s2 + style statistics + LeakyReLU + instance norm) -- torch autograd on
    oracle.ref_nets.discr_block against central differences of oracle.np_nets.discr_block (no autograd involved), for a scalar
    that uses BOTH outputs: entries of every parameter tensor and of the input, 1e-6 relative."""
    from oracle import np_nets as NN
    from oracle import ref_nets as R
    x, ws, head, rng = _block_case(100)
    cy = rng.normal(size=(2, 4, 4, 4))

    def scalar_np(xx, ww):
        y, st = NN.discr_block(xx, *ww)
        return float((y * cy).sum() + (st @ head).sum())

    xt = torch.tensor(x, requires_grad=True)
    wt = [torch.tensor(w, requires_grad=True) for w in ws]
    y, st = R.discr_block(xt, *wt, return_styles=True)
    ((y * torch.tensor(cy)).sum() + (st @ torch.tensor(head)).sum()).backward()
    assert abs(scalar_np(x, ws) - float((y * torch.tensor(cy)).sum() + (st @ torch.tensor(head)).sum())) < 1e-10
    for ti, w in enumerate(ws):
        for _ in range(4):
            idx = tuple(int(rng.integers(0, d)) for d in w.shape)
            fd = NN.central_difference(lambda v, ti=ti: scalar_np(x, ws[:ti] + [v] + ws[ti + 1:]), w, idx, 1e-4)
            an = float(wt[ti].grad[idx])
            assert abs(fd - an) <= 1e-6 * max(1.0, abs(an)), ("parameter", ti, idx, fd, an)
    for _ in range(6):
        idx = tuple(int(rng.integers(0, d)) for d in x.shape)
        fd = NN.central_difference(lambda v: scalar_np(v, ws), x, idx, 1e-4)
        an = float(xt.grad[idx])
        assert abs(fd - an) <= 1e-6 * max(1.0, abs(an)), ("input", idx, fd, an)


def test_r1_penalty_double_backward_against_nested_finite_differences():
    """The gradient-of-gradient behind gradient_regularization (losses.py:75-82): oracle.ref_ops.r1_penalty differentiates
    |d out / d x|^2 w.r.t. the weights through torch's double backward.  Here the inner gradient is taken by central differences
    of the NumPy DiscrBlock + style head over EVERY input element, the outer derivative by central differences of that -- no
    autograd anywhere -- and compared at 1e-5 relative (nested differences: ~1e-7 of noise)."""
    from oracle import np_nets as NN
    from oracle import ref_nets as R
    from oracle import ref_ops as O
    x, ws, head, rng = _block_case(300)
    hb = rng.normal(size=1)

    def out_np(xx, ww):                                  # (N,) head output of the block's styles
        return (NN.discr_block(xx, *ww)[1] @ head + hb)[:, 0]

    def penalty_np(ww, h=1e-5):
        g = np.zeros_like(x)
        xx = x.copy()
        for idx in np.ndindex(*x.shape):
            old = xx[idx]
            xx[idx] = old + h
            fp = out_np(xx, ww).sum()
            xx[idx] = old - h
            fm = out_np(xx, ww).sum()
            xx[idx] = old
            g[idx] = (fp - fm) / (2 * h)
        return 10 * 0.5 * (g.reshape(g.shape[0], -1) ** 2).sum(axis=1).mean()

    xt = torch.tensor(x, requires_grad=True)
    wt = [torch.tensor(w, requires_grad=True) for w in ws]
    st = R.discr_block(xt, *wt, return_styles=True)[1]
    pen = O.r1_penalty(st @ torch.tensor(head) + torch.tensor(hb), xt)
    assert abs(float(pen) - penalty_np(ws)) <= 1e-7 * max(1.0, float(pen))
    grads = torch.autograd.grad(pen, wt[:2])
    for ti in (0, 1):                                    # conv kernel and bias (gamma / beta do not reach the pre-activation styles)
        for _ in range(3):
            idx = tuple(int(rng.integers(0, d)) for d in ws[ti].shape)
            fd = NN.central_difference(lambda v, ti=ti: penalty_np(ws[:ti] + [v] + ws[ti + 1:]), ws[ti], idx, 2e-3)
            an = float(grads[ti][idx])
            assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (ti, idx, fd, an)


def test_rotation_gradient_through_the_interpolation_weights_against_finite_differences():
    """transform_3d_grid_tf (confignet_utils.py:100-159) is differentiable w.r.t. the rotation only through the interpolation
    weights (the cell indices come from floor()): d/d angles of a scalar of the resampled volume -- torch autograd through
    oracle.ref_ops.transform_3d_grid and euler_angles_to_matrix -- equals central differences of the per-voxel NumPy loop
    (no voxel changes its cell within the step; clamped coordinates carry no gradient either way)."""
    from oracle import np_nets as NN
    rng = np.random.default_rng(8)
    grid = rng.normal(size=(1, 4, 4, 4, 3))
    ang = np.array([[0.37, -0.21, 0.11]])
    cw = rng.normal(size=grid.shape)

    def scalar_np(a):
        return float((NP.transform_3d_grid(grid, NP.euler_angles_to_matrix(a)) * cw).sum())

    at = torch.tensor(ang, requires_grad=True)
    gt = torch.tensor(grid, requires_grad=True)
    out = O.transform_3d_grid(gt, O.euler_angles_to_matrix(at))
    (out * torch.tensor(cw)).sum().backward()
    assert abs(scalar_np(ang) - float((out * torch.tensor(cw)).sum())) < 1e-10
    for k in range(3):
        fd = NN.central_difference(scalar_np, ang, (0, k), 1e-6)
        assert abs(fd - float(at.grad[0, k])) <= 1e-6 * max(1.0, abs(float(at.grad[0, k]))), (k, fd, float(at.grad[0, k]))
    # the volume itself enters linearly: its gradient is the transposed interpolation
    idx = (0, 1, 2, 1, 0)
    g2 = grid.copy()
    g2[idx] += 1.0
    lin = float((NP.transform_3d_grid(g2, NP.euler_angles_to_matrix(ang)) * cw).sum()) - scalar_np(ang)
    assert abs(lin - float(gt.grad[idx])) < 1e-10
