"""Timed CPU baseline for bench.py: the oracle's restatement of ONE whole second-stage training
iteration (torch-CPU fp32, eager, op for op with the reference) on a bounded batch.
TEST/BENCH INFRASTRUCTURE (oracle/__init__.py): this is a port of the reference's arithmetic, not
TensorFlow 2.1 (which cannot be installed here)."""
import os
import time
from collections import OrderedDict

import numpy as np
import torch

from . import ref_nets as R
from . import ref_ops as O
from . import ref_steps as S

FACEMODEL_IO = OrderedDict(sorted({
    "beard_style_embedding": (9, 7), "blendshape_values": (62, 30), "bone_rotations:left_eye": (3, 2),
    "eye_color": (8, 3), "eyebrow_style_embedding": (44, 7), "geometry_identity_params": (53, 30),
    "hdri_embedding": (50, 20), "head_hair_color": (3, 3), "head_hair_style_embedding": (18, 9),
    "lower_eyelash_style": (4, 2), "texture_embedding": (50, 30), "upper_eyelash_style": (4, 2)}.items()))


def _init(shapes, rng, he=False, dtype=torch.float32, grad=True):
    out = []
    for s in shapes:
        if len(s) == 1:
            a = np.zeros(s, np.float32)
        elif he:
            rf = int(np.prod(s[:-2])) if len(s) > 2 else 1
            a = (rng.standard_normal(size=s) * np.sqrt(2.0 / (rf * s[-2]))).astype(np.float32)
        else:
            a = O.glorot_uniform(rng, s)
        out.append(torch.tensor(a, dtype=dtype, requires_grad=grad))
    return out


def build_weights(res, latent_dim, rng, dtype=torch.float32):
    W = {
        "generator": _init(R.generator_weight_shapes(latent_dim, res), rng, dtype=dtype),
        "discriminator": _init(R.discriminator_weight_shapes(res), rng, dtype=dtype),
        "synth_discriminator": _init(R.discriminator_weight_shapes(res), rng, dtype=dtype),
        "latent_discriminator": _init(R.mlp_weight_shapes(4, latent_dim, latent_dim, 1), rng, dtype=dtype),
        "latent_regressor": _init(R.latent_regressor_weight_shapes(latent_dim, res), rng, dtype=dtype),
        "synthetic_encoder": _init(R.synthetic_encoder_weight_shapes(list(FACEMODEL_IO.values())), rng, dtype=dtype),
    }
    with torch.no_grad():
        W["generator"][1].fill_(1.0)           # learned_input bias = ones
        W["generator"][0].zero_()
        for k in ("discriminator", "synth_discriminator", "latent_regressor"):
            shapes = R.discriminator_weight_shapes(res) if k != "latent_regressor" else R.latent_regressor_weight_shapes(latent_dim, res)
            for i in range(5):
                W[k][2 + 4 * i + 2].fill_(1.0)  # instance-norm gamma = 1
    ew = _init(R.real_encoder_weight_shapes(latent_dim), rng, he=True, dtype=dtype)
    with torch.no_grad():
        for w, role in zip(ew, R.resnet50_weight_roles()):       # (the 4 head tensors after the ResNet keep their init)
            if role in ("gamma", "var"):
                w.fill_(1.0)
            if role in ("mean", "var"):
                w.requires_grad_(False)
    W["real_encoder"] = ew
    W["generator_smoothed"] = [w.detach().clone() for w in W["generator"]]
    vgg = _init(R.vgg_weight_shapes(R.VGG19_CFG), rng, he=True, dtype=dtype, grad=False)
    return W, vgg


def make_batch(res, b, rng, dtype=torch.float32):
    t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
    img = lambda n: t(rng.uniform(-1, 1, size=(n, res, res, 3)))
    par = lambda n: [t(rng.standard_normal(size=(n, d[0]))) for d in FACEMODEL_IO.values()]

    def rot(n):
        r = np.zeros((n, 3))
        r[:, 0] = np.pi * rng.uniform(-30, 30, n) / 180
        r[:, 1] = np.pi * rng.uniform(-10, 10, n) / 180
        return t(r)
    ns, nr = b // 2, b - b // 2
    masks = np.zeros((ns, res, res), np.uint8)
    masks[:, res // 3:res // 3 + res // 12, res // 3:res // 3 + res // 12] = 1
    return {"real_d": img(b), "enc_in_d": img(b), "real_sd": img(b), "params_sd": par(b), "rot_sd": rot(b),
            "real_ld": img(b), "params_ld": par(b), "params_g": par(ns), "rot_g": rot(ns), "synth_imgs_g": img(ns),
            "eye_masks_g": torch.as_tensor(masks), "real_imgs_g": img(nr)}


STATE_NETS = ("generator", "generator_smoothed", "discriminator", "synth_discriminator", "latent_discriminator",
              "latent_regressor", "synthetic_encoder", "real_encoder")
STATE_D_NETS = ("discriminator", "synth_discriminator", "latent_discriminator")


def flip_subset(imgs, flags):
    """flip_random_subset_of_images (confignet_utils.py:24-31): images with a set flag are mirrored left-right."""
    out = np.array(imgs, copy=True)
    for i, f in enumerate(flags):
        if f:
            out[i] = out[i, :, ::-1]
    return out


def load_state(path, dtype=torch.float32):
    """The HIP path's own weights and batches of ONE iteration, written by bench.py:dump_parity_state (a flat .npz): the
    iteration below is then the checker of that iteration's loss scalars (bench line: loss_parity_vs_cpu).
    Returns (W, vgg, batch, post_d): post_d = the device path's discriminator weights after its discriminator phase,
    handed over before the generator step exactly as tests/test_steps_gpu.py does (lr*sign(g) steps of noise-level
    gradient entries differ between any two fp32 summation orders)."""
    z = np.load(path)
    t = lambda a, grad=False: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=grad)
    W = {}
    for net in STATE_NETS:
        n = int(z["n/" + net])
        W[net] = [t(z["%s/%d" % (net, i)], net != "generator_smoothed") for i in range(n)]
    for w, role in zip(W["real_encoder"], R.resnet50_weight_roles()):
        if role in ("mean", "var"):
            w.requires_grad_(False)
    vgg = [t(z["vgg/%d" % i]) for i in range(int(z["n/vgg"]))]
    post_d = {net: [t(z["post/%s/%d" % (net, i)]) for i in range(int(z["n/" + net]))] for net in STATE_D_NETS}
    names = list(FACEMODEL_IO.keys())
    img = lambda key, flip=None: t((flip_subset(z[key], z[flip]) if flip else z[key]).astype(np.float64) / 127.5 - 1.0)
    batch = {"real_d": img("img/real_d", "flip/real_d"), "enc_in_d": img("img/enc_in_d"),
             "real_sd": img("img/real_sd", "flip/real_sd"), "params_sd": [t(z["sd/p/" + n]) for n in names], "rot_sd": t(z["sd/rot"]),
             "real_ld": img("img/real_ld", "flip/real_ld"), "params_ld": [t(z["ld/p/" + n]) for n in names],
             "params_g": [t(z["g/p/" + n]) for n in names], "rot_g": t(z["g/rot"]), "synth_imgs_g": img("img/synth_g"),
             "eye_masks_g": torch.as_tensor(z["eye_masks_g"]), "real_imgs_g": img("img/real_g", "flip/real_g")}
    return W, vgg, batch, post_d


def time_second_stage_iteration(res=256, batch=2, repeats=1, threads=None, warmup=0, state=None, probe_threads=(), probe_limit_s=25):
    """Returns (images_per_sec, median seconds per iteration, threads used[, loss dicts of the state iteration]).
    state: path of a bench.py parity dump -- its weights replace the seeded ones and its batch is the FIRST iteration run
    (with fresh Adam moments, as the device path ran it); that iteration's four loss dicts are returned as floats."""
    threads = threads or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    rng = np.random.default_rng(0)
    latent_dim = sum(v[1] for v in FACEMODEL_IO.values())
    cfg = {"output_shape": (res, res, 3), "rotation_ranges": ((-30, 30), (-10, 10), (0, 0)),
           "image_loss_weight": 5e-4, "eye_loss_weight": 5, "domain_adverserial_loss_weight": 5.0,
           "latent_regression_weight": 10.0, "latent_regressor_rot_weight": 5.0}
    parity = None
    if state is None:
        W, vgg = build_weights(res, latent_dim, rng)
        first = post_d = None
    else:
        W, vgg, first, post_d = load_state(state)
        batch = first["real_d"].shape[0]

    def hand_over(Wd):
        with torch.no_grad():
            for net in STATE_D_NETS:
                for w, a in zip(Wd[net], post_d[net]):
                    w.copy_(a)
    d_opt = O.KerasAdam(lr=4e-4, beta_1=0.0, beta_2=0.9)
    g_opt = O.KerasAdam(lr=4e-4, beta_1=0.0, beta_2=0.9)
    times = []
    probe = {}
    for i in range(warmup + repeats):
        if i == warmup and probe_threads:
            # Thread-count probe, after the warm-up iteration has paid the one-off costs (40 s against 8 s per iteration measured):
            # one iteration AT THIS BATCH per candidate, threads pinned by the caller's OMP_PLACES / OMP_PROC_BIND, each under a
            # SIGALRM limit (torch-CPU eager collapses from oversubscription somewhere beyond a few dozen threads on a 256-core
            # host: an all-cores iteration did not finish in 14 minutes).  The fastest count times the remaining iterations.
            import signal
            best, best_sec = threads, float("inf")
            for c in probe_threads:
                torch.set_num_threads(c)
                signal.signal(signal.SIGALRM, _alarm)
                signal.alarm(int(probe_limit_s))
                t0 = time.perf_counter()
                try:
                    S.second_stage_iteration(W, cfg, make_batch(res, batch, rng), d_opt, g_opt, vgg)
                    sec_c = time.perf_counter() - t0
                    probe[c] = round(sec_c, 2)
                    if sec_c < best_sec:
                        best, best_sec = c, sec_c
                except _Timeout:
                    probe[c] = "> %d s" % probe_limit_s
                finally:
                    signal.alarm(0)
            threads = best
            torch.set_num_threads(threads)
        use_state = first is not None and i == 0
        b = first if use_state else make_batch(res, batch, rng)
        t0 = time.perf_counter()
        out = S.second_stage_iteration(W, cfg, b, d_opt, g_opt, vgg, after_discriminator_phase=hand_over if use_state else None)
        if i >= warmup:
            times.append(time.perf_counter() - t0)
        if use_state:
            parity = {step: {k: float(v.detach()) for k, v in d.items()} for step, d in out.items()}
    sec = float(np.median(times)) if times else float("nan")
    if probe_threads:
        time_second_stage_iteration.last_probe = probe
    if state is not None:
        return batch / sec, sec, threads, parity
    return batch / sec, sec, threads


class _Timeout(Exception):
    pass


def _alarm(signum, frame):
    raise _Timeout()


if __name__ == "__main__":
    import json
    import sys
    # usage: python -m oracle.cpu_baseline BATCH RES [STATE.npz [--parity-only]]
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    r = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    state = sys.argv[3] if len(sys.argv) > 3 else None
    host = os.cpu_count() or 1
    if "--parity-only" in sys.argv:                 # tests: the state iteration alone, no timing
        _, _, cores, parity = time_second_stage_iteration(r, b, repeats=0, threads=min(16, host), warmup=1, state=state)
        print(json.dumps({"cores": cores, "parity_losses": parity}))
        sys.exit(0)
    cands = sorted({min(c, host) for c in (16, 32, 64, 128)})
    # 1 warm-up at 16 threads (= the parity iteration, if any), the thread-count probe, then the median of 3 at the fastest count
    res_ = time_second_stage_iteration(r, b, repeats=3, threads=min(16, host), warmup=1, state=state, probe_threads=cands)
    v, sec, cores = res_[:3]
    print(json.dumps({"value": v, "seconds": sec, "cores": cores, "host_cores": host,
                      "thread_probe_seconds": getattr(time_second_stage_iteration, "last_probe", {}),
                      "parity_losses": res_[3] if state is not None else None}))
