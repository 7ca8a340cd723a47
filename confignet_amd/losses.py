"""Losses (reference: confignet/losses.py) on HIP kernels."""
import os

import numpy as np
import torch

from . import functional as F

BATCHED_R1 = os.environ.get("CN_NO_BATCHED_R1") is None
BATCHED_TANGENT = os.environ.get("CN_NO_BATCHED_TANGENT") is None     # the six heads' tangent passes as one stacked pass


def _r1_penalties(discriminator, out_real, real_imgs, inter):
    """{gp_loss_i} of losses.py:75-82 for every head, without a second-order tape (see compute_discriminator_loss)."""
    from . import ops
    if BATCHED_R1 and BATCHED_TANGENT and hasattr(discriminator, "tangent_all"):
        g_img = discriminator.input_gradients(inter, stacked=True).detach()      # all heads in one tape-free backward sweep
        jvps = discriminator.tangent_all(g_img, inter)                           # ... and one stacked tangent pass
        n = real_imgs.shape[0]
        sq = ops.row_sumsq(g_img.reshape(g_img.shape[0], -1))                    # |g_i,n|^2 (constant of the tape)
        return {"gp_loss_" + str(i): 10 * 0.5 * (2.0 * jvp.reshape(-1) - sq[i * n:(i + 1) * n]).mean() for i, jvp in enumerate(jvps)}
    if BATCHED_R1 and hasattr(discriminator, "input_gradients"):
        gs = discriminator.input_gradients(inter)              # all six heads in one tape-free backward sweep
    else:
        with F.input_grads_only():
            gs = [torch.autograd.grad(o, real_imgs, grad_outputs=torch.ones_like(o), retain_graph=True)[0]
                  for o in out_real.values()]
    gp = {}
    for i, g in enumerate(gs):
        g = g.detach()
        jvp = discriminator.tangent(g, inter, i).reshape(-1)          # == |g_n|^2, carries d/dtheta
        gp["gp_loss_" + str(i)] = 10 * 0.5 * (2.0 * jvp - F.row_sumsq(g)).mean()
    return gp


def GAN_G_loss(scores):
    """mean(softplus(-scores)) (losses.py:7-8)."""
    return F.gan_loss(scores, 1.0)


def GAN_D_loss(labels, scores):
    """mean(labels*softplus(-s) + (1-labels)*softplus(s)) (losses.py:10-11).  `labels` is a python
    scalar or an array of 0/1 values (the reference only ever passes those)."""
    if np.isscalar(labels):
        return F.gan_loss(scores, float(labels))
    lab = np.asarray(labels).reshape(-1)
    vals = np.unique(lab)
    if len(vals) == 1:
        return F.gan_loss(scores, float(vals[0]))
    total = 0
    for v in vals:
        idx = torch.as_tensor(np.nonzero(lab == v)[0], device=scores.device)
        total = total + F.gan_loss(scores.reshape(-1)[idx].reshape(-1, 1), float(v)) * (len(idx) / len(lab))
    return total


def eye_loss(gt_imgs, gen_imgs, eye_masks):
    """mean_n( sum_hwc ((gt-gen)*mask)^2 / (1 + sum_hw mask) ) (losses.py:13-18); masks uint8 (N,H,W)."""
    if not torch.is_tensor(eye_masks):
        eye_masks = torch.as_tensor(np.ascontiguousarray(eye_masks)).to(gen_imgs.device)
    diff = F.MaskedDiffFn.apply(gen_imgs, gt_imgs, eye_masks.contiguous())
    den = 1.0 + eye_masks.reshape(eye_masks.shape[0], -1).sum(dim=1).to(torch.float32)
    return (F.row_sumsq(diff) / den).mean()


def gradient_regularization(real_out, real_in):
    """R1 (losses.py:75-82): 10*0.5*mean_n sum (d sum(real_out) / d real_in)^2; the input-gradient pass
    is itself recorded (create_graph) so the penalty can be differentiated w.r.t. the weights."""
    with F.input_grads_only():
        (g,) = torch.autograd.grad(real_out, real_in, grad_outputs=torch.ones_like(real_out), create_graph=True)
    return 10 * 0.5 * F.row_sumsq(g).mean()


def compute_discriminator_loss(discriminator, real_imgs, fake_imgs, second_order_tape=False):
    """losses.py:20-47: GAN loss on real and fake for each of the 6 heads + R1 penalty per head.

    Default: no second-order tape.  g_i = d sum(out_i)/d real is taken by an ordinary (fused, first-order)
    backward pass; since d/dtheta |g_i|^2 = 2 d/dtheta JVP_x(out_i)(v)|_{v = g_i}, the penalty is expressed as
    2*JVP - |v|^2 (equal in value to |g_i|^2) where the JVP is a tangent forward pass of the discriminator
    (HologanDiscriminator.tangent).  One first-order backward then yields exactly the gradient
    tf.GradientTape's nested tapes produce.  second_order_tape=True keeps the literal reverse-over-reverse
    formulation on twice-differentiable composite ops (used as the cross-check in the tests)."""
    if second_order_tape:
        return _compute_discriminator_loss_tape(discriminator, real_imgs, fake_imgs)
    real_imgs = real_imgs.detach().requires_grad_(True)
    inter = []
    out_real = discriminator(real_imgs, intermediates=inter)
    out_fake = discriminator(fake_imgs.detach())
    losses = {}
    for i, o in enumerate(out_real.values()):
        losses["GAN_loss_real_" + str(i)] = GAN_D_loss(1.0, o)
    for i, o in enumerate(out_fake.values()):
        losses["GAN_loss_fake_" + str(i)] = GAN_D_loss(0.0, o)
    losses.update(_r1_penalties(discriminator, out_real, real_imgs, inter))
    losses["loss_sum"] = sum(losses.values())
    return losses


def discriminator_loss_real(discriminator, real_imgs):
    """The terms of compute_discriminator_loss that depend on the real images only: GAN_loss_real_i and gp_loss_i."""
    real_imgs = real_imgs.detach().requires_grad_(True)
    inter = []
    out_real = discriminator(real_imgs, intermediates=inter)
    real = {}
    for i, o in enumerate(out_real.values()):
        real["GAN_loss_real_" + str(i)] = GAN_D_loss(1.0, o)
    return real, _r1_penalties(discriminator, out_real, real_imgs, inter)


def discriminator_loss_fake(discriminator, fake_imgs):
    out_fake = discriminator(fake_imgs.detach())
    return {"GAN_loss_fake_" + str(i): GAN_D_loss(0.0, o) for i, o in enumerate(out_fake.values())}


def _compute_discriminator_loss_tape(discriminator, real_imgs, fake_imgs):
    real_imgs = real_imgs.detach().requires_grad_(True)
    out_real = discriminator(real_imgs, twice_differentiable=True)
    out_fake = discriminator(fake_imgs.detach())
    losses = {}
    for i, o in enumerate(out_real.values()):
        losses["GAN_loss_real_" + str(i)] = GAN_D_loss(1.0, o)
    for i, o in enumerate(out_fake.values()):
        losses["GAN_loss_fake_" + str(i)] = GAN_D_loss(0.0, o)
    for i, o in enumerate(out_real.values()):
        losses["gp_loss_" + str(i)] = gradient_regularization(o, real_imgs)
    losses["loss_sum"] = sum(losses.values())
    return losses


def compute_latent_discriminator_loss(latent_discriminator, real_latents, fake_latents):
    """losses.py:49-73."""
    real_latents = real_latents.detach().requires_grad_(True)
    out_real = latent_discriminator(real_latents, twice_differentiable=True)
    out_fake = latent_discriminator(fake_latents.detach())
    losses = {
        "GAN_loss_real": GAN_D_loss(1.0, out_real),
        "GAN_loss_fake": GAN_D_loss(0.0, out_fake),
        "gp_loss": gradient_regularization(out_real, real_latents),
    }
    losses["loss_sum"] = sum(losses.values())
    return losses


def mean_squared_error(labels, outputs):
    """reduce_mean(tf.losses.mean_squared_error(a, b)) == global mean (R7).  Only ever applied to
    (N, latent_dim+3) tensors (<= a few thousand floats, gradients on BOTH sides through the encoders):
    latent-vector algebra of this size is host-side plumbing, not a kernel."""
    return ((labels - outputs) ** 2).mean()


def compute_latent_regression_loss(generator_outputs, labels, latent_regressor):
    """losses.py:85-90."""
    return mean_squared_error(labels, latent_regressor(generator_outputs))
