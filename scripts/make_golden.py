"""Generates tests/golden/first_stage_128.npz from the CPU oracle (float64): seeded Keras-ordered weights and
inputs -> generator image crop + checksum, the 6 discriminator logits, every loss scalar of the first-stage
steps (BASELINE.json configs[0] run at 128x128), and a Keras-Adam trace.

The reference itself cannot be executed here (tensorflow-gpu==2.1.0 is not installable, no weights), so these
vectors pin the ORACLE (regression) and give the GPU tests a fixture that needs no oracle run; they are not
TensorFlow outputs."""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_nets as R, ref_ops as O, ref_steps as S   # noqa: E402

FM = OrderedDict([("beard_style_embedding", (9, 7)), ("blendshape_values", (62, 30)), ("eye_color", (8, 3)),
                  ("head_hair_color", (3, 3))])
RES, L = 128, 43


def seeded_weights(shapes, seed, he=False):
    rng = np.random.default_rng(seed)
    out = []
    for s in shapes:
        if len(s) == 1:
            out.append((rng.standard_normal(s) * 0.05).astype(np.float32))
        elif he:
            rf = int(np.prod(s[:-2])) if len(s) > 2 else 1
            out.append((rng.standard_normal(s) * np.sqrt(2.0 / (rf * s[-2]))).astype(np.float32))
        else:
            out.append(O.glorot_uniform(rng, s))
    return out


def build():
    W = {
        "generator": seeded_weights(R.generator_weight_shapes(L, RES), 1),
        "discriminator": seeded_weights(R.discriminator_weight_shapes(RES), 2),
        "synth_discriminator": seeded_weights(R.discriminator_weight_shapes(RES), 3),
        "latent_discriminator": seeded_weights(R.mlp_weight_shapes(4, L, L, 1), 4),
        "latent_regressor": seeded_weights(R.latent_regressor_weight_shapes(L, RES), 5),
        "synthetic_encoder": seeded_weights(R.synthetic_encoder_weight_shapes(list(FM.values())), 6),
    }
    W["generator"][0][:] = 0.0
    W["generator"][1][:] = 1.0 + W["generator"][1] * 0.0
    for k in ("discriminator", "synth_discriminator", "latent_regressor"):
        for i in range(5):
            W[k][2 + 4 * i + 2] = (1.0 + W[k][2 + 4 * i + 2]).astype(np.float32)     # instance-norm gamma ~ 1
    vgg = seeded_weights(R.vgg_weight_shapes(R.VGG19_CFG), 7, he=True)
    rng = np.random.default_rng(8)
    n = 2
    inp = {
        "z": rng.standard_normal((n, L)), "rot": np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.17, 0.17, n), np.zeros(n)], 1),
        "real": rng.uniform(-1, 1, (n, RES, RES, 3)), "fake": rng.uniform(-1, 1, (n, RES, RES, 3)),
        "params": [rng.standard_normal((1, d[0])) for d in FM.values()],
        "synth_rot": np.array([[0.2, -0.1, 0.0]]), "gt": rng.uniform(-1, 1, (1, RES, RES, 3)),
        "z_real": rng.standard_normal((1, L)), "rot_real": np.array([[-0.3, 0.05, 0.0]]),
    }
    masks = np.zeros((1, RES, RES), np.uint8)
    masks[:, 40:52, 50:70] = 1
    inp["masks"] = masks
    return W, vgg, inp


def t64(a, grad=False):
    return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=grad)


def compute(W, vgg, inp):
    cfg = {"output_shape": (RES, RES, 3), "image_loss_weight": 5e-5, "eye_loss_weight": 5, "domain_adverserial_loss_weight": 5.0,
           "latent_regression_weight": 10.0, "latent_regressor_rot_weight": 5.0}
    Wt = {k: [t64(w, True) for w in v] for k, v in W.items()}
    vt = [t64(w) for w in vgg]
    out = {}
    img = R.generator_forward(Wt["generator"], t64(inp["z"]), t64(inp["rot"]), RES)
    out["gen_crop"] = img[:, 48:80, 48:80, :].detach().numpy()
    out["gen_checksum"] = np.array([float(img.sum()), float((img ** 2).sum())])
    logits = R.discriminator_forward(Wt["discriminator"], t64(inp["real"]))
    out["d_logits"] = np.concatenate([v.detach().numpy() for v in logits.values()], axis=1)
    dl = S.discriminator_loss(Wt["discriminator"], t64(inp["real"]), t64(inp["fake"]))
    out["d_loss_names"] = np.array(list(dl.keys()))
    out["d_loss_values"] = np.array([float(v) for v in dl.values()])
    ld = S.latent_discriminator_loss(Wt["latent_discriminator"], t64(inp["z"]), t64(inp["z"][::-1].copy() * 0.5))
    out["ld_loss_names"] = np.array(list(ld.keys()))
    out["ld_loss_values"] = np.array([float(v) for v in ld.values()])
    gl, _ = S.first_stage_generator_loss(Wt, cfg, [t64(p) for p in inp["params"]], t64(inp["synth_rot"]), t64(inp["gt"]),
                                         torch.as_tensor(inp["masks"]), t64(inp["z_real"]), t64(inp["rot_real"]), vt)
    out["g_loss_names"] = np.array(list(gl.keys()))
    out["g_loss_values"] = np.array([float(v) for v in gl.values()])
    grads = S.grads_of(gl["loss_sum"], Wt["generator"])
    out["g_grad_norms"] = np.array([float(g.norm()) for g in grads])
    # Keras Adam with the shared counter: 4 applications on a 1000-element tensor
    rng = np.random.default_rng(9)
    th, g = rng.standard_normal(1000), rng.standard_normal(1000)
    p = t64(th).clone()
    opt = O.KerasAdam(lr=4e-4, beta_1=0.0, beta_2=0.9)
    trace = []
    for t in range(1, 5):
        opt.apply_gradients([(t64(g * t), p)])
        trace.append(p.numpy().copy())
    out["adam_theta0"], out["adam_grad"], out["adam_trace"] = th, g, np.stack(trace)
    return out


def build_second_stage():
    """Second-stage extras (confignet_second_stage.py:149-218): seeded real-encoder (ResNet-50 v1 + two heads) weights,
    a real image for the encoder branch; the rest is shared with the first-stage fixture."""
    W, vgg, inp = build()
    enc = seeded_weights(R.real_encoder_weight_shapes(L), 10, he=True)
    rng = np.random.default_rng(11)
    # BatchNorm statistics near the identity so that 50 layers neither vanish nor explode
    for i, role in enumerate(R.resnet50_weight_roles()):
        c = enc[i].shape[0]
        if role == "bias":
            enc[i] = (0.05 * rng.standard_normal(c)).astype(np.float32)
        elif role == "gamma":
            enc[i] = (0.5 + 0.05 * rng.standard_normal(c)).astype(np.float32)      # (< 1: keeps 16 residual blocks O(1))
        elif role in ("beta", "mean"):
            enc[i] = (0.05 * rng.standard_normal(c)).astype(np.float32)
        elif role == "var":
            enc[i] = (1.0 + 0.05 * np.abs(rng.standard_normal(c))).astype(np.float32)
    enc[-4] = (enc[-4] * 0.05).astype(np.float32)      # heads: keep tanh unsaturated and the latents O(1)
    enc[-2] = (enc[-2] * 0.05).astype(np.float32)
    W["real_encoder"] = enc
    inp["real_for_encoder"] = rng.uniform(-1, 1, (1, RES, RES, 3))
    return W, vgg, inp


def compute_second_stage(W, vgg, inp):
    cfg = {"output_shape": (RES, RES, 3), "image_loss_weight": 5e-4, "eye_loss_weight": 5, "domain_adverserial_loss_weight": 5.0,
           "latent_regression_weight": 10.0, "latent_regressor_rot_weight": 5.0, "rotation_ranges": ((-30, 30), (-10, 10), (0, 0))}
    Wt = {k: [t64(w, True) for w in v] for k, v in W.items()}
    vt = [t64(w) for w in vgg]
    lat, rot = R.real_encoder_forward(Wt["real_encoder"], t64(inp["real_for_encoder"]), cfg["rotation_ranges"])
    gl, _ = S.second_stage_generator_loss(Wt, cfg, [t64(p) for p in inp["params"]], t64(inp["synth_rot"]), t64(inp["gt"]),
                                          torch.as_tensor(inp["masks"]), t64(inp["real_for_encoder"]), vt)
    grads = S.grads_of(gl["loss_sum"], Wt["real_encoder"])
    return {"enc_latents": lat.detach().numpy(), "enc_rotations": rot.detach().numpy(),
            "g_loss_names": np.array(list(gl.keys())), "g_loss_values": np.array([float(v) for v in gl.values()]),
            "enc_grad_norm_total": np.array([float(torch.sqrt(sum((g ** 2).sum() for g in grads if g is not None)))])}


# ---- fine_tune_on_img (confignet_second_stage.py:321-403; BASELINE.json configs[3] at 128x128) -----------------------
FT_EXPR = (7, 37)        # blendshape_values slice of the 43-d latent: after the 7-d beard_style_embedding (sorted keys)


def build_fine_tune(n_imgs=1):
    W, vgg, inp = build_second_stage()
    # a trained learned_input is not the all-ones initial bias: with a constant 4^3 input most first-layer channels see one sign
    # only, their bias gradient is exactly zero in exact arithmetic and fp32 noise then decides the sign of an lr*sign(g) step
    W["generator"][1] = (1.0 + 0.5 * np.random.default_rng(14).standard_normal(32768)).astype(np.float32)
    W["generator_smoothed"] = [w.copy() for w in W["generator"]]
    vggface = seeded_weights(R.vgg_weight_shapes(R.VGG16_CFG), 12, he=True)
    rng = np.random.default_rng(13)
    inp["ft_imgs"] = rng.uniform(-1, 1, (n_imgs, RES, RES, 3))
    return W, vgg, vggface, inp


FT_CFG = {"output_shape": (RES, RES, 3), "image_loss_weight": 5e-4, "domain_adverserial_loss_weight": 5.0,
          "latent_regression_weight": 10.0, "latent_regressor_rot_weight": 5.0,
          "rotation_ranges": ((-30, 30), (-10, 10), (0, 0))}


def compute_fine_tune(W, vgg, vggface, inp, n_iters=3):
    Wt = {k: [t64(w) for w in v] for k, v in W.items()}
    emb0, rot0 = R.real_encoder_forward(Wt["real_encoder"], t64(inp["ft_imgs"]), FT_CFG["rotation_ranges"])
    emb, rot, hist, gen = S.fine_tune_on_img(Wt, FT_CFG, t64(inp["ft_imgs"]), n_iters, [t64(w) for w in vgg],
                                             [t64(w) for w in vggface], FT_EXPR)
    emb1, rot1, _, _ = S.fine_tune_on_img(Wt, FT_CFG, t64(inp["ft_imgs"]), 1, [t64(w) for w in vgg],
                                          [t64(w) for w in vggface], FT_EXPR)
    return {"loss_names": np.array(list(hist[0].keys())),
            "loss_values": np.array([[h[k] for k in hist[0].keys()] for h in hist]),
            "emb_encoder": emb0.numpy(), "rot_encoder": rot0.numpy(), "emb": emb.numpy(), "rot": rot.numpy(),
            "emb_1iter": emb1.numpy(), "rot_1iter": rot1.numpy(),
            "gen_delta_norms": np.array([float((a - t64(b)).norm()) for a, b in zip(gen, W["generator_smoothed"])])}


if __name__ == "__main__":
    W, vgg, inp = build()
    out = compute(W, vgg, inp)
    path = os.path.join(ROOT, "tests", "golden", "first_stage_128.npz")
    if "--second-only" not in sys.argv:
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes")
    for k in ("d_loss_values", "g_loss_values", "gen_checksum"):
        print(k, out[k])
    W2, vgg2, inp2 = build_second_stage()
    out2 = compute_second_stage(W2, vgg2, inp2)
    path2 = os.path.join(ROOT, "tests", "golden", "second_stage_128.npz")
    np.savez_compressed(path2, **out2)
    print("wrote", path2, os.path.getsize(path2), "bytes")
    for k in ("g_loss_names", "g_loss_values", "enc_rotations", "enc_grad_norm_total"):
        print(k, out2[k])
    Wf, vggf, vggfacef, inpf = build_fine_tune()
    out3 = compute_fine_tune(Wf, vggf, vggfacef, inpf)
    path3 = os.path.join(ROOT, "tests", "golden", "fine_tune_128.npz")
    np.savez_compressed(path3, **out3)
    print("wrote", path3, os.path.getsize(path3), "bytes")
    for k in ("loss_names", "loss_values", "rot_encoder", "rot", "rot_1iter"):
        print(k, out3[k])
