"""Golden fixtures (tests/golden/first_stage_128.npz, made by scripts/make_golden.py from the float64 oracle).
CPU: the oracle still reproduces them (regression pin).  GPU: the HIP path matches them at the north_star
tolerance (1e-3 max-abs on outputs / loss scalars) WITHOUT running the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import make_golden as MG   # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "first_stage_128.npz"))


def test_oracle_reproduces_golden_generator_and_discriminator():
    W, vgg, inp = MG.build()
    Wt = {k: [MG.t64(w) for w in v] for k, v in W.items()}
    from oracle import ref_nets as R
    img = R.generator_forward(Wt["generator"], MG.t64(inp["z"]), MG.t64(inp["rot"]), MG.RES)
    np.testing.assert_allclose(img[:, 48:80, 48:80, :].numpy(), GOLD["gen_crop"], atol=1e-12)
    np.testing.assert_allclose([float(img.sum()), float((img ** 2).sum())], GOLD["gen_checksum"], rtol=1e-12)
    logits = R.discriminator_forward(Wt["discriminator"], MG.t64(inp["real"]))
    np.testing.assert_allclose(np.concatenate([v.numpy() for v in logits.values()], axis=1), GOLD["d_logits"], atol=1e-12)
    assert list(GOLD["d_loss_names"]) == ["GAN_loss_real_%d" % i for i in range(6)] + ["GAN_loss_fake_%d" % i for i in range(6)] + \
        ["gp_loss_%d" % i for i in range(6)] + ["loss_sum"]


def test_oracle_adam_trace_matches_golden():
    from oracle import ref_ops as O
    p = MG.t64(GOLD["adam_theta0"]).clone()
    opt = O.KerasAdam(lr=4e-4, beta_1=0.0, beta_2=0.9)
    for t in range(1, 5):
        opt.apply_gradients([(MG.t64(GOLD["adam_grad"] * t), p)])
        np.testing.assert_allclose(p.numpy(), GOLD["adam_trace"][t - 1], atol=1e-15)


def _model():
    from confignet_amd import ConfigNetFirstStage
    W, vgg, inp = MG.build()
    cfg = {"output_shape": (MG.RES, MG.RES, 3), "batch_size": 2, "facemodel_inputs": dict(MG.FM)}
    m = ConfigNetFirstStage(cfg, seed=0)
    assert m.config["latent_dim"] == MG.L
    m.generator.set_weights(W["generator"]); m.generator_smoothed.set_weights(W["generator"])
    m.discriminator.set_weights(W["discriminator"]); m.synth_discriminator.set_weights(W["synth_discriminator"])
    m.latent_discriminator.set_weights(W["latent_discriminator"]); m.latent_regressor.set_weights(W["latent_regressor"])
    m.synthetic_encoder.set_weights(W["synthetic_encoder"])
    m.perceptual_loss._pretrained_dnn_activations.set_weights(vgg)
    return m, inp


@pytest.mark.gpu
def test_hip_path_matches_golden_first_stage():
    from confignet_amd.confignet_first_stage import frozen
    from confignet_amd.losses import compute_discriminator_loss, compute_latent_discriminator_loss
    m, inp = _model()
    dev = m._dev
    img = m.generator((inp["z"], inp["rot"].astype(np.float32)))
    got = img.detach().cpu().double().numpy()
    assert np.abs(got[:, 48:80, 48:80, :] - GOLD["gen_crop"]).max() < 1e-3
    np.testing.assert_allclose([got.sum(), (got ** 2).sum()], GOLD["gen_checksum"], rtol=1e-4)
    logits = m.discriminator(inp["real"])
    assert np.abs(np.concatenate([v.detach().cpu().numpy() for v in logits.values()], axis=1) - GOLD["d_logits"]).max() < 1e-3
    dl = compute_discriminator_loss(m.discriminator, dev(inp["real"]), dev(inp["fake"]))
    assert list(dl.keys()) == list(GOLD["d_loss_names"])
    for k, v in zip(dl.keys(), GOLD["d_loss_values"]):
        assert abs(float(dl[k]) - v) <= 1e-3 * max(1.0, abs(v)), (k, float(dl[k]), v)
    ld = compute_latent_discriminator_loss(m.latent_discriminator, dev(inp["z"]), dev(inp["z"][::-1].copy() * 0.5))
    for k, v in zip(ld.keys(), GOLD["ld_loss_values"]):
        assert abs(float(ld[k]) - v) <= 1e-3 * max(1.0, abs(v)), (k, float(ld[k]), v)
    with frozen(m.discriminator, m.synth_discriminator, m.latent_discriminator):
        gl = m._generator_loss([dev(p) for p in inp["params"]], dev(inp["synth_rot"]), dev(inp["gt"]),
                               torch.as_tensor(inp["masks"]).cuda(), dev(inp["z_real"]), dev(inp["rot_real"]))
        grads = torch.autograd.grad(gl["loss_sum"], m.generator.trainable_weights)
    assert list(gl.keys()) == list(GOLD["g_loss_names"])
    for k, v in zip(gl.keys(), GOLD["g_loss_values"]):
        assert abs(float(gl[k]) - v) <= 1e-3 * max(1.0, abs(v)), (k, float(gl[k]), v)
    norms = np.array([float(g.norm()) for g in grads])
    np.testing.assert_allclose(norms, GOLD["g_grad_norms"], rtol=2e-2, atol=1e-6)     # fp32 ReLU-branch noise, see test_nets_gpu


@pytest.mark.gpu
def test_hip_adam_matches_golden_trace():
    from confignet_amd import ops
    import math
    th = torch.tensor(GOLD["adam_theta0"], dtype=torch.float32).cuda()
    m_, v_ = torch.zeros_like(th), torch.zeros_like(th)
    for t in range(1, 5):
        lr_t = torch.tensor([4e-4 * math.sqrt(1 - 0.9 ** t)], dtype=torch.float32).cuda()
        ops.adam_step(th, torch.tensor(GOLD["adam_grad"] * t, dtype=torch.float32).cuda(), m_, v_, None, lr_t, 0.0, 0.9, 1e-7)
        assert np.abs(th.cpu().double().numpy() - GOLD["adam_trace"][t - 1]).max() < 2e-6


GOLD2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "second_stage_128.npz"))


def test_second_stage_golden_is_reproduced_by_the_oracle_encoder():
    """(CPU) the committed encoder outputs come from scripts/make_golden.py's seeded ResNet-50 restatement."""
    from oracle import ref_nets as R
    W, vgg, inp = MG.build_second_stage()
    lat, rot = R.real_encoder_forward([torch.tensor(w, dtype=torch.float64) for w in W["real_encoder"]],
                                      torch.tensor(inp["real_for_encoder"]))
    assert np.abs(lat.numpy() - GOLD2["enc_latents"]).max() < 1e-9 and np.abs(rot.numpy() - GOLD2["enc_rotations"]).max() < 1e-12


@pytest.mark.gpu
def test_hip_path_matches_golden_second_stage():
    """ConfigNet (second stage) generator step on the HIP path against the committed oracle vector: real encoder
    (ResNet-50, BatchNorm in inference mode) outputs and all 18 loss scalars incl. the batch-normalised latent regression."""
    from confignet_amd import ConfigNet
    from confignet_amd.confignet_first_stage import frozen
    W, vgg, inp = MG.build_second_stage()
    cfg = {"output_shape": (MG.RES, MG.RES, 3), "batch_size": 2, "facemodel_inputs": dict(MG.FM)}
    m = ConfigNet(cfg, seed=0)
    m.config["image_loss_weight"] *= 10                                    # train_confignet.py:67
    m.generator.set_weights(W["generator"]); m.discriminator.set_weights(W["discriminator"])
    m.synth_discriminator.set_weights(W["synth_discriminator"]); m.latent_discriminator.set_weights(W["latent_discriminator"])
    m.latent_regressor.set_weights(W["latent_regressor"]); m.synthetic_encoder.set_weights(W["synthetic_encoder"])
    m.encoder.set_weights(W["real_encoder"])
    m.perceptual_loss._pretrained_dnn_activations.set_weights(vgg)
    dev = m._dev
    lat, rot = m.encoder(dev(inp["real_for_encoder"]))
    assert np.abs(lat.detach().cpu().double().numpy() - GOLD2["enc_latents"]).max() < 1e-3
    assert np.abs(rot.detach().cpu().double().numpy() - GOLD2["enc_rotations"]).max() < 1e-4
    with frozen(m.discriminator, m.synth_discriminator, m.latent_discriminator):
        gl = m._generator_loss([dev(p) for p in inp["params"]], dev(inp["synth_rot"]), dev(inp["gt"]),
                               torch.as_tensor(inp["masks"]).cuda(), dev(inp["real_for_encoder"]))
        grads = torch.autograd.grad(gl["loss_sum"], m.encoder.trainable_weights)
    assert list(gl.keys()) == list(GOLD2["g_loss_names"])
    for k, v in zip(gl.keys(), GOLD2["g_loss_values"]):
        assert abs(float(gl[k]) - v) <= 1e-3 * max(1.0, abs(v)), (k, float(gl[k]), v)
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
    assert abs(total - float(GOLD2["enc_grad_norm_total"][0])) <= 3e-2 * float(GOLD2["enc_grad_norm_total"][0])


GOLD3 = np.load(os.path.join(os.path.dirname(__file__), "golden", "fine_tune_128.npz"))


def test_oracle_reproduces_fine_tune_golden_and_the_reference_fixture_facts():
    """(CPU) tests/golden/fine_tune_128.npz comes from oracle/ref_steps.py:fine_tune_on_img.  The facts the reference's own
    confignet_finetune_ref_*.npz show without weights hold for it: after one iteration exactly the 30
    blendshape dims (7..36) of the embedding moved, each by lr = 1e-4, and each rotation moved by 1e-4."""
    W, vgg, vggface, inp = MG.build_fine_tune()
    out = MG.compute_fine_tune(W, vgg, vggface, inp, n_iters=2)
    np.testing.assert_allclose(out["loss_values"], GOLD3["loss_values"][:2], rtol=1e-9)
    np.testing.assert_allclose(out["emb_1iter"], GOLD3["emb_1iter"], atol=1e-12)
    d = (GOLD3["emb_1iter"] - GOLD3["emb_encoder"])[0]
    assert np.array_equal(np.nonzero(d)[0], np.arange(7, 37))
    assert np.all(np.abs(np.abs(d[7:37]) - 1e-4) < 1e-7)
    assert np.all(np.abs(np.abs((GOLD3["rot_1iter"] - GOLD3["rot_encoder"])[0]) - 1e-4) < 1e-7)
    assert list(GOLD3["loss_names"]) == ["image_loss_real", "face_reco_loss"] + ["GAN_loss_real_%d" % i for i in range(6)] + \
        ["latent_GAN_loss", "latent_regression_loss", "loss_sum"]


def test_resnet50_weight_order_is_keras_layer_order():
    """keras.applications ResNet50 lists a conv-shortcut block as 1_conv, 1_bn, 2_conv, 2_bn, 0_conv, 3_conv, 0_bn, 3_bn
    (depth-sorted functional-model layers, [TF-2.1]); get_weights() follows it with [kernel, bias] per Conv2D and
    [gamma, beta, moving_mean, moving_variance] per BatchNormalization.  Product and oracle must agree on it, and the
    shapes of the first conv-shortcut block are pinned here."""
    from oracle import ref_nets as R
    from confignet_amd.dnn_models.real_encoder import resnet50_layers
    prod = list(resnet50_layers())
    assert prod == R.resnet50_layer_order()
    names = [n for n, *_ in prod]
    assert names[:2] == ["conv1_conv", "conv1_bn"]
    assert names[2:10] == ["conv2_block1_1_conv", "conv2_block1_1_bn", "conv2_block1_2_conv", "conv2_block1_2_bn",
                           "conv2_block1_0_conv", "conv2_block1_3_conv", "conv2_block1_0_bn", "conv2_block1_3_bn"]
    assert names[10:16] == ["conv2_block2_1_conv", "conv2_block2_1_bn", "conv2_block2_2_conv", "conv2_block2_2_bn",
                            "conv2_block2_3_conv", "conv2_block2_3_bn"]
    assert len(names) == 2 * 53 and names[-1] == "conv5_block3_3_bn"
    shp = R.resnet50_weight_shapes()
    assert len(shp) == 53 * 6 and shp[:2] == [(7, 7, 3, 64), (64,)]
    # conv2_block1: ..., 2_bn x4, then 0_conv (1,1,64,256) BEFORE 3_conv (1,1,64,256), then 0_bn x4, 3_bn x4
    assert shp[6:8] == [(1, 1, 64, 64), (64,)] and shp[12:14] == [(3, 3, 64, 64), (64,)]
    assert shp[18:22] == [(1, 1, 64, 256), (256,), (1, 1, 64, 256), (256,)] and shp[22:30] == [(256,)] * 8
    # conv3_block1 distinguishes the two by shape: 0_conv reads the 256-channel block input, 3_conv the 128-channel branch
    i = names.index("conv3_block1_0_conv")
    convs_before = sum(1 for n in names[:i] if n.endswith("_conv"))
    off = 2 * convs_before + 4 * (i - convs_before)
    assert shp[off] == (1, 1, 256, 512) and shp[off + 2] == (1, 1, 128, 512)
    assert sum(int(np.prod(s)) for s in shp) == 23587712           # keras ResNet50(include_top=False) parameter count
