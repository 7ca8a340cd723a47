"""Oracle training steps: restatement of the reference's step functions on explicit
batches (the numpy batch sampling is done by the caller so the HIP path and the oracle
see identical inputs).  TEST INFRASTRUCTURE (oracle/__init__.py).

Weights are dicts name -> list of torch tensors (requires_grad=True leaves), names as in
ConfigNetFirstStage.get_weights() minus the "_weights" suffix.
"""
import torch

from . import ref_ops as O
from . import ref_nets as R


def discriminator_loss(d_w, real_imgs, fake_imgs):
    """compute_discriminator_loss (losses.py:20-47)."""
    real_imgs = real_imgs.detach().requires_grad_(True)
    out_real = R.discriminator_forward(d_w, real_imgs)
    out_fake = R.discriminator_forward(d_w, fake_imgs.detach())
    losses = {}
    ones = torch.ones(real_imgs.shape[0], 1, dtype=real_imgs.dtype)
    zeros = torch.zeros(fake_imgs.shape[0], 1, dtype=real_imgs.dtype)
    for i, o in enumerate(out_real.values()):
        losses["GAN_loss_real_%d" % i] = O.gan_d_loss(ones, o)
    for i, o in enumerate(out_fake.values()):
        losses["GAN_loss_fake_%d" % i] = O.gan_d_loss(zeros, o)
    for i, o in enumerate(out_real.values()):
        losses["gp_loss_%d" % i] = O.r1_penalty(o, real_imgs)
    losses["loss_sum"] = sum(losses.values())
    return losses


def latent_discriminator_loss(ld_w, real_latents, fake_latents):
    """compute_latent_discriminator_loss (losses.py:49-73); MLP with LeakyReLU(0.3)."""
    real_latents = real_latents.detach().requires_grad_(True)
    out_real = O.mlp_simple(real_latents, ld_w, 0.3)
    out_fake = O.mlp_simple(fake_latents.detach(), ld_w, 0.3)
    n = real_latents.shape[0]
    losses = {
        "GAN_loss_real": O.gan_d_loss(torch.ones(n, 1, dtype=out_real.dtype), out_real),
        "GAN_loss_fake": O.gan_d_loss(torch.zeros(n, 1, dtype=out_real.dtype), out_fake),
        "gp_loss": O.r1_penalty(out_real, real_latents),
    }
    losses["loss_sum"] = sum(losses.values())
    return losses


def grads_of(loss, weight_list):
    gs = torch.autograd.grad(loss, weight_list, allow_unused=True)
    return [torch.zeros_like(w) if g is None else g for g, w in zip(gs, weight_list)]


def first_stage_generator_loss(W, cfg, facemodel_params, synth_rot, gt_imgs, eye_masks,
                               real_latents, real_rot, vgg_w):
    """ConfigNetFirstStage.generator_training_step (confignet_first_stage.py:506-560),
    the part inside the tape.  gt_imgs already scaled to [-1, 1]."""
    res = cfg["output_shape"][0]
    losses = {}
    synth_latents = R.synthetic_encoder_forward(W["synthetic_encoder"], facemodel_params)
    g_synth = R.generator_forward(W["generator"], synth_latents, synth_rot, res)
    g_real = R.generator_forward(W["generator"], real_latents, real_rot, res)
    losses["image_loss"] = cfg["image_loss_weight"] * R.perceptual_loss(vgg_w, gt_imgs, g_synth)
    losses["eye_loss"] = cfg["eye_loss_weight"] * O.eye_loss(gt_imgs, g_synth, eye_masks)
    for i, o in enumerate(R.discriminator_forward(W["synth_discriminator"], g_synth).values()):
        losses["GAN_loss_synth_%d" % i] = O.gan_g_loss(o)
    for i, o in enumerate(R.discriminator_forward(W["discriminator"], g_real).values()):
        losses["GAN_loss_real_%d" % i] = O.gan_g_loss(o)
    ld_out = O.mlp_simple(synth_latents, W["latent_discriminator"], 0.3)
    losses["latent_GAN_loss"] = cfg["domain_adverserial_loss_weight"] * O.gan_g_loss(ld_out)
    lat = torch.cat([synth_latents, real_latents], dim=0)
    imgs = torch.cat([g_synth, g_real], dim=0)
    rots = torch.cat([synth_rot, real_rot], dim=0)
    labels = torch.cat([lat, cfg["latent_regressor_rot_weight"] * rots], dim=-1)
    pred = R.latent_regressor_forward(W["latent_regressor"], imgs)
    # tf.losses.mean_squared_error + reduce_mean == global mean (R7)
    losses["latent_regression_loss"] = cfg["latent_regression_weight"] * ((labels - pred) ** 2).mean()
    losses["loss_sum"] = sum(losses.values())
    return losses, {"g_synth": g_synth, "g_real": g_real, "synth_latents": synth_latents}


def normalized_latent_regression_loss(lr_w, cfg, gen_imgs, labels):
    """ConfigNet.compute_normalized_latent_regression_loss (confignet_second_stage.py:93-107)."""
    out = R.latent_regressor_forward(lr_w, gen_imgs)
    den = torch.sqrt(labels.var(dim=0, unbiased=False, keepdim=True) + 1e-3)
    den = torch.cat([den[:, :-3], torch.ones(1, 3, dtype=den.dtype)], dim=1)
    out = out.mean(dim=0) + (out - out.mean(dim=0)) / den
    labels = labels.mean(dim=0) + (labels - labels.mean(dim=0)) / den
    return ((labels - out) ** 2).mean() * cfg["latent_regression_weight"]


def second_stage_generator_loss(W, cfg, facemodel_params, synth_rot, synth_imgs, eye_masks,
                                real_imgs, vgg_w):
    """ConfigNet.generator_training_step (confignet_second_stage.py:149-218), inside the tape."""
    res = cfg["output_shape"][0]
    n_synth, n_real = synth_imgs.shape[0], real_imgs.shape[0]
    losses = {}
    synth_latents = R.synthetic_encoder_forward(W["synthetic_encoder"], facemodel_params)
    g_synth = R.generator_forward(W["generator"], synth_latents, synth_rot, res)
    real_latents, real_rot = R.real_encoder_forward(W["real_encoder"], real_imgs, cfg["rotation_ranges"])
    g_real = R.generator_forward(W["generator"], real_latents, real_rot, res)
    losses["image_loss_synth"] = cfg["image_loss_weight"] * R.perceptual_loss(vgg_w, synth_imgs, g_synth)
    losses["image_loss_real"] = cfg["image_loss_weight"] * R.perceptual_loss(vgg_w, real_imgs, g_real)
    losses["eye_loss"] = cfg["eye_loss_weight"] * O.eye_loss(synth_imgs, g_synth, eye_masks)
    for i, o in enumerate(R.discriminator_forward(W["synth_discriminator"], g_synth).values()):
        losses["GAN_loss_synth_%d" % i] = O.gan_g_loss(o)
    for i, o in enumerate(R.discriminator_forward(W["discriminator"], g_real).values()):
        losses["GAN_loss_real_%d" % i] = O.gan_g_loss(o)
    ld_synth = O.mlp_simple(synth_latents, W["latent_discriminator"], 0.3)
    ld_real = O.mlp_simple(real_latents, W["latent_discriminator"], 0.3)
    ld_out = torch.cat([ld_real, ld_synth], dim=0)
    dt = ld_out.dtype
    dom_labels = torch.cat([torch.zeros(n_real, 1, dtype=dt), torch.ones(n_synth, 1, dtype=dt)], dim=0)
    losses["latent_GAN_loss"] = cfg["domain_adverserial_loss_weight"] * O.gan_d_loss(dom_labels, ld_out)
    if cfg["latent_regression_weight"] > 0.0:
        lat = torch.cat([synth_latents, real_latents], dim=0)
        imgs = torch.cat([g_synth, g_real], dim=0)
        rots = torch.cat([synth_rot, real_rot], dim=0)
        labels = torch.cat([lat, cfg["latent_regressor_rot_weight"] * rots], dim=-1)
        losses["latent_regression_loss"] = normalized_latent_regression_loss(
            W["latent_regressor"], cfg, imgs, labels)
    losses["loss_sum"] = sum(losses.values())
    return losses, {"g_synth": g_synth, "g_real": g_real}


def ema_update(smoothed, training, alpha=0.999):
    """update_smoothed_weights (confignet_first_stage.py:393-400)."""
    with torch.no_grad():
        for s, t in zip(smoothed, training):
            s.mul_(alpha).add_(t, alpha=1 - alpha)


def second_stage_iteration(W, cfg, batch, d_opt, g_opt, vgg_w, keep_grads=None, after_discriminator_phase=None):
    """One whole reference training iteration (confignet_second_stage.py:277-288) on explicit
    batches: D step, synth-D step, latent-D step, G step, EMA.  `batch` holds, for a batch size B:
    real_d, enc_in_d (B images each), real_sd, params_sd, rot_sd, real_ld, params_ld, and for the G step
    params_g, rot_g, synth_imgs_g, eye_masks_g (B//2) and real_imgs_g (B - B//2).
    Used by tests and as bench.py's timed CPU baseline.  keep_grads: optional dict that receives the gradient lists
    of the four steps (tests compare the Adam updates where the gradient is significant).  after_discriminator_phase(W):
    optional hook between step (3) and step (4) (a test checks the three discriminator updates there and then continues
    from the device path's post-update discriminator weights, so that fp32 sign flips of lr*sign(g) steps on noise-level
    gradients are not amplified into the generator step's loss scalars)."""
    res = cfg["output_shape"][0]
    out = {}
    kg = keep_grads if keep_grads is not None else {}
    # (1) discriminator step (confignet_first_stage.py:466-476 ; batch: second_stage:119-130)
    with torch.no_grad():
        lat, rot = R.real_encoder_forward(W["real_encoder"], batch["enc_in_d"], cfg["rotation_ranges"])
        fake = R.generator_forward(W["generator"], lat, rot, res)
    losses = discriminator_loss(W["discriminator"], batch["real_d"], fake)
    kg["discriminator"] = grads_of(losses["loss_sum"], W["discriminator"])
    d_opt.apply_gradients(list(zip(kg["discriminator"], W["discriminator"])))
    out["d"] = losses
    # (2) synthetic-domain discriminator step (confignet_first_stage.py:478-488,452-464)
    with torch.no_grad():
        lat = R.synthetic_encoder_forward(W["synthetic_encoder"], batch["params_sd"])
        fake = R.generator_forward(W["generator"], lat, batch["rot_sd"], res)
    losses = discriminator_loss(W["synth_discriminator"], batch["real_sd"], fake)
    kg["synth_discriminator"] = grads_of(losses["loss_sum"], W["synth_discriminator"])
    d_opt.apply_gradients(list(zip(kg["synth_discriminator"], W["synth_discriminator"])))
    out["synth_d"] = losses
    # (3) latent discriminator step (confignet_second_stage.py:132-147)
    with torch.no_grad():
        real_lat, _ = R.real_encoder_forward(W["real_encoder"], batch["real_ld"], cfg["rotation_ranges"])
        fake_lat = R.synthetic_encoder_forward(W["synthetic_encoder"], batch["params_ld"])
    losses = latent_discriminator_loss(W["latent_discriminator"], real_lat, fake_lat)
    kg["latent_discriminator"] = grads_of(losses["loss_sum"], W["latent_discriminator"])
    d_opt.apply_gradients(list(zip(kg["latent_discriminator"], W["latent_discriminator"])))
    out["latent_d"] = losses
    if after_discriminator_phase is not None:
        after_discriminator_phase(W)
    # (4) generator step (confignet_second_stage.py:149-218)
    losses, _ = second_stage_generator_loss(W, cfg, batch["params_g"], batch["rot_g"], batch["synth_imgs_g"],
                                            batch["eye_masks_g"], batch["real_imgs_g"], vgg_w)
    allw = W["generator"] + W["latent_regressor"] + W["synthetic_encoder"] + \
        [w for w in W["real_encoder"] if w.requires_grad]
    kg["g_step"] = grads_of(losses["loss_sum"], allw)
    g_opt.apply_gradients(list(zip(kg["g_step"], allw)))
    out["g"] = losses
    # (5) EMA (confignet_first_stage.py:393-400)
    ema_update(W["generator_smoothed"], W["generator"])
    return out


def fine_tune_on_img(W, cfg, input_images, n_iters, vgg_w, vggface_w, expr_slice, force_neutral_expression=False,
                     neutral_expr_latents=None, lr=0.0001):
    """ConfigNet.fine_tune_on_img (confignet_second_stage.py:321-403) on explicit weights.

    W: generator_smoothed (copied into the fine-tuned generator, l.337), real_encoder, discriminator,
    latent_discriminator, latent_regressor.  input_images already scaled to [-1, 1], (n_imgs, R, R, 3).
    expr_slice = (start, stop) of the blendshape_values slice of the latent (l.340).
    neutral_expr_latents: (1, stop-start) latents of the all-zero blendshape vector (l.328-331) when
    force_neutral_expression.  Returns (embeddings, rotations, per-step loss dicts, fine-tuned generator weights);
    the embeddings are the pre/post tiles computed BEFORE the last optimizer step with the updated expr (l.402)."""
    res = cfg["output_shape"][0]
    dt = input_images.dtype
    with torch.no_grad():
        emb, rot = R.real_encoder_forward(W["real_encoder"], input_images, cfg["rotation_ranges"])     # l.327
        emb = emb.clone()
        if force_neutral_expression:
            emb[:, expr_slice[0]:expr_slice[1]] = neutral_expr_latents.to(dt)
    gen = [w.detach().clone().requires_grad_(True) for w in W["generator_smoothed"]]                   # l.333-337
    mean_emb = emb.mean(dim=0, keepdim=True)                                                           # l.341
    pre = mean_emb[:, :expr_slice[0]].clone().requires_grad_(True)                                     # l.343
    expr = emb[:, expr_slice[0]:expr_slice[1]].clone().requires_grad_(not force_neutral_expression)    # l.344
    post = mean_emb[:, expr_slice[1]:].clone().requires_grad_(True)                                    # l.345
    rotations = rot.clone().requires_grad_(True)                                                       # l.348
    n_imgs = input_images.shape[0]
    opt = O.KerasAdam(lr=lr)                                                                           # l.350: default betas 0.9 / 0.999
    history = []
    pre_t = post_t = None
    for _ in range(n_iters):
        losses = {}
        pre_t, post_t = pre.repeat(n_imgs, 1), post.repeat(n_imgs, 1)                                  # l.363-364
        embeddings = torch.cat([pre_t, expr, post_t], dim=1)                                           # l.366
        out = R.generator_forward(gen, embeddings, rotations, res)                                     # l.368
        losses["image_loss_real"] = 0.5 * cfg["image_loss_weight"] * R.perceptual_loss(vgg_w, input_images, out)          # l.369
        losses["face_reco_loss"] = 0.5 * cfg["image_loss_weight"] * R.perceptual_loss(vggface_w, out, input_images, "VGGFace")   # l.370,88-91
        for i, o in enumerate(R.discriminator_forward(W["discriminator"], out).values()):              # l.373-376
            losses["GAN_loss_real_%d" % i] = O.gan_g_loss(o)
        ld_out = O.mlp_simple(embeddings, W["latent_discriminator"], 0.3)                              # l.379
        ones = torch.ones(1, 1, dtype=dt)                                                              # fake_y_real, l.351
        losses["latent_GAN_loss"] = cfg["domain_adverserial_loss_weight"] * O.gan_d_loss(ones, ld_out)  # l.380-382
        labels = torch.cat([embeddings, cfg["latent_regressor_rot_weight"] * rotations], dim=-1)       # l.385
        losses["latent_regression_loss"] = normalized_latent_regression_loss(W["latent_regressor"], cfg, out, labels)   # l.388
        losses["loss_sum"] = sum(losses.values())                                                      # l.390
        tw = gen + [pre, post, rotations] + ([] if force_neutral_expression else [expr])               # l.392-394
        grads = torch.autograd.grad(losses["loss_sum"], tw, allow_unused=True)                         # l.395
        opt.apply_gradients(list(zip(grads, tw)))                                                      # l.396
        history.append({k: float(v.detach()) for k, v in losses.items()})
    final = torch.cat([pre_t, expr, post_t], dim=1)                                                    # l.402: stale tiles, fresh expr
    return final.detach(), rotations.detach(), history, [g.detach() for g in gen]


def latent_gan_generator_step(g_w, d_w, z, opt):
    """LatentGAN.generator_training_step (latent_gan.py:151-165): generator MLP (LeakyReLU 0.3) on z, G loss through the
    discriminator, Adam on the generator's weights."""
    fake = O.mlp_simple(z, g_w, 0.3)
    out = O.mlp_simple(fake, d_w, 0.3)
    loss = O.gan_g_loss(out)
    opt.apply_gradients(list(zip(grads_of(loss, g_w), g_w)))
    return {"gan_loss": loss, "loss_sum": loss}


def latent_gan_discriminator_step(g_w, d_w, real_latents, z, opt):
    """LatentGAN.discriminator_training_step (latent_gan.py:117-149): fakes from generator.predict(z) (outside the tape)."""
    with torch.no_grad():
        fake = O.mlp_simple(z, g_w, 0.3)
    losses = latent_discriminator_loss(d_w, real_latents, fake)
    opt.apply_gradients(list(zip(grads_of(losses["loss_sum"], d_w), d_w)))
    return losses
