"""Losses (reference: confignet/losses.py) on HIP kernels."""

import numpy as np
import torch

from . import functional as F

BATCHED_R1 = True
BATCHED_TANGENT = True     # the six heads' tangent passes as one stacked pass


def _r1_penalties(discriminator, out_real, real_imgs, inter):
    """{gp_loss_i} of losses.py:75-82 for every head, without a second-order tape (see compute_discriminator_loss)."""
    from . import ops
    if BATCHED_R1 and BATCHED_TANGENT and hasattr(discriminator, "tangent_all"):
        g_img = discriminator.input_gradients(inter, stacked=True).detach()      # all heads in one tape-free backward sweep
        jvps = discriminator.tangent_all(g_img, inter)                           # ... and one stacked tangent pass
        n = real_imgs.shape[0]
        sq = ops.row_sumsq(g_img.reshape(g_img.shape[0], -1))                    # |g_i,n|^2 (constant of the tape)
        # all heads at once: 10 * 0.5 * mean_n(2 jvp - |g|^2) per head (five small launches instead of five per head)
        gp = (5.0 * (2.0 * torch.cat([j.reshape(1, n) for j in jvps], dim=0) - sq.reshape(len(jvps), n))).mean(dim=1)
        return {"gp_loss_" + str(i): gp[i] for i in range(len(jvps))}
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


def total(values):
    """sum of the scalar loss terms (the reference's `sum(losses.values())`, losses.py:45, confignet_first_stage.py:552): one
    stack + one reduction instead of a chain of ~20 two-operand adds (each a launch of its own, forward and backward)."""
    vals = [v.reshape(()).float() if torch.is_tensor(v) else torch.as_tensor(float(v)) for v in values]
    dev = next((v.device for v in vals if v.is_cuda), None)
    if dev is not None:
        vals = [v if v.is_cuda else v.to(dev) for v in vals]
    return torch.stack(vals).sum() if len(vals) > 1 else vals[0]


def GAN_G_loss(scores):
    """mean(softplus(-scores)) (losses.py:7-8)."""
    return F.gan_loss(scores, 1.0)


def GAN_G_losses(score_tensors):
    """[GAN_G_loss(s) for s in score_tensors] -- the heads of one discriminator call -- from one launch (F.gan_losses)."""
    score_tensors = list(score_tensors)
    return F.gan_losses(score_tensors, [1.0] * len(score_tensors))


def GAN_D_losses(label, score_tensors):
    """[GAN_D_loss(label, s) for s in score_tensors] for one constant label, from one launch."""
    score_tensors = list(score_tensors)
    return F.gan_losses(score_tensors, [float(label)] * len(score_tensors))


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
    for i, l in enumerate(GAN_D_losses(1.0, out_real.values())):
        losses["GAN_loss_real_" + str(i)] = l
    for i, l in enumerate(GAN_D_losses(0.0, out_fake.values())):
        losses["GAN_loss_fake_" + str(i)] = l
    losses.update(_r1_penalties(discriminator, out_real, real_imgs, inter))
    losses["loss_sum"] = total(losses.values())
    return losses


def discriminator_loss_real(discriminator, real_imgs):
    """The terms of compute_discriminator_loss that depend on the real images only: GAN_loss_real_i and gp_loss_i."""
    real_imgs = real_imgs.detach().requires_grad_(True)
    inter = []
    out_real = discriminator(real_imgs, intermediates=inter)
    real = {"GAN_loss_real_" + str(i): l for i, l in enumerate(GAN_D_losses(1.0, out_real.values()))}
    return real, _r1_penalties(discriminator, out_real, real_imgs, inter)


def discriminator_loss_fake(discriminator, fake_imgs):
    out_fake = discriminator(fake_imgs.detach())
    return {"GAN_loss_fake_" + str(i): l for i, l in enumerate(GAN_D_losses(0.0, out_fake.values()))}


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
    losses["loss_sum"] = total(losses.values())
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
    losses["loss_sum"] = total(losses.values())
    return losses


class GlobalBatchMoments(torch.autograd.Function):
    """(mean, biased variance) over the GLOBAL batch of an (N_local, F) tensor under data parallelism: the per-feature sums
    and sums of squares are added over the ranks (2 F floats), and in the backward pass the cotangents of the two statistics
    are added over the ranks before they are pushed to the local rows -- each rank's loss depends on every rank's rows through
    the statistics, and the data-parallel objective is the mean of the ranks' losses (option (i) of SURVEY.md section 8e for
    confignet_second_stage.py:93-107; equal shards assumed, as everywhere in the data-parallel path)."""

    @staticmethod
    def forward(ctx, x):
        from . import parallel
        n = x.shape[0] * parallel.world_size()
        s = torch.cat((x.sum(dim=0), (x * x).sum(dim=0)))
        parallel.allreduce_sum_inline(s)
        f = x.shape[1]
        mean = s[:f] / n
        var = s[f:] / n - mean * mean
        ctx.save_for_backward(x, mean)
        ctx.n = n
        return mean, var

    @staticmethod
    def backward(ctx, g_mean, g_var):
        from . import parallel
        x, mean = ctx.saved_tensors
        f = x.shape[1]
        g = torch.cat((g_mean if g_mean is not None else torch.zeros_like(mean), g_var if g_var is not None else torch.zeros_like(mean))).contiguous()
        parallel.allreduce_sum_inline(g)
        return (g[:f] + 2.0 * g[f:] * (x - mean)) / ctx.n


class DeferredGlobalStatsRegression:
    """normalized_latent_regression with GLOBAL batch statistics in a form whose collectives all run on the calling thread
    (a captured step cuts its HIP graph at a collective, and a graph cannot be cut from autograd's worker thread, where a
    Function's backward would run).  Forward: the three statistics come from one all-reduce of the local sums and enter the
    loss as LEAVES.  `cotangents()` -- called by the step's update, before the main backward pass -- takes the loss's gradient
    w.r.t. (labels, out, leaves) through the few nodes of the normalisation, adds the leaves' gradients over the ranks and folds
    them back into the cotangents of labels / out; the main backward pass then continues from (labels, out) with those
    cotangents next to the rest of the loss (`.loss` itself must stay OUT of the differentiated sum: use `.value`)."""

    def __init__(self, out, labels, weight):
        from . import parallel
        self.out, self.labels = out, labels
        f = labels.shape[1]
        self.n = labels.shape[0] * parallel.world_size()
        with torch.no_grad():
            s = torch.cat((labels.sum(dim=0), (labels * labels).sum(dim=0), out.sum(dim=0))).contiguous()
        parallel.allreduce_sum_inline(s)
        with torch.no_grad():
            l_mean = s[:f] / self.n
            l_var = s[f:2 * f] / self.n - l_mean * l_mean
            o_mean = s[2 * f:] / self.n
        self.leaves = [t.clone().requires_grad_(True) for t in (l_mean, l_var, o_mean)]
        lm, lv, om = (t.unsqueeze(0) for t in self.leaves)
        den = torch.sqrt(lv + 1e-3)
        den = torch.cat((den[:, :-3], torch.ones((1, 3), device=den.device, dtype=den.dtype)), dim=1)
        self.loss = mean_squared_error(lm + (labels - lm) / den, om + (out - om) / den) * weight
        self.value = self.loss.detach()

    def cotangents(self):
        """[(tensor, cotangent)] for labels and out: what the main backward pass continues from."""
        from . import parallel
        srcs = [t for t in (self.labels, self.out) if t.requires_grad]
        grads = torch.autograd.grad(self.loss, srcs + self.leaves, allow_unused=True)
        g_src = dict(zip([id(t) for t in srcs], grads[:len(srcs)]))
        g = torch.cat([gi if gi is not None else torch.zeros_like(l) for gi, l in zip(grads[len(srcs):], self.leaves)]).contiguous()
        parallel.allreduce_sum_inline(g)
        f = self.labels.shape[1]
        g_lm, g_lv, g_om = g[:f], g[f:2 * f], g[2 * f:]
        out = []
        if self.labels.requires_grad:
            with torch.no_grad():
                extra = (g_lm + 2.0 * g_lv * (self.labels - self.leaves[0])) / self.n
            out.append((self.labels, g_src[id(self.labels)] + extra))
        if self.out.requires_grad:
            out.append((self.out, g_src[id(self.out)] + g_om / self.n))
        return out


def normalized_latent_regression(out, labels, weight, global_statistics=False):
    """The arithmetic of ConfigNet.compute_normalized_latent_regression_loss (confignet_second_stage.py:96-107) on the
    regressor's output: both sides are re-centred and the latent part is divided by the batch standard deviation of the labels.
    global_statistics: the batch statistics are those of the GLOBAL batch of a data-parallel step (GlobalBatchMoments), so
    that W ranks at batch B/W optimise exactly the single-process objective at batch B."""
    if global_statistics:
        l_mean, l_var = GlobalBatchMoments.apply(labels)
        o_mean, _ = GlobalBatchMoments.apply(out)
        l_mean, l_var, o_mean = l_mean.unsqueeze(0), l_var.unsqueeze(0), o_mean.unsqueeze(0)
    else:
        l_mean, l_var = labels.mean(dim=0, keepdim=True), labels.var(dim=0, unbiased=False, keepdim=True)
        o_mean = out.mean(dim=0, keepdim=True)
    den = torch.sqrt(l_var + 1e-3)
    den = torch.cat((den[:, :-3], torch.ones((1, 3), device=den.device, dtype=den.dtype)), dim=1)
    out = o_mean + (out - o_mean) / den
    labels = l_mean + (labels - l_mean) / den
    return mean_squared_error(labels, out) * weight


def mean_squared_error(labels, outputs):
    """reduce_mean(tf.losses.mean_squared_error(a, b)) == global mean (R7).  Only ever applied to
    (N, latent_dim+3) tensors (<= a few thousand floats, gradients on BOTH sides through the encoders):
    latent-vector algebra of this size is host-side plumbing, not a kernel."""
    return ((labels - outputs) ** 2).mean()


def compute_latent_regression_loss(generator_outputs, labels, latent_regressor):
    """losses.py:85-90."""
    return mean_squared_error(labels, latent_regressor(generator_outputs))
