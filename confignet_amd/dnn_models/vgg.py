"""VGG-19 / VGG-16 convolutional slices up to block4_conv2 (keras.applications definitions used by
confignet/perceptual_loss.py:19-41) on HIP kernels.  The weights are frozen (no filter gradient).
Offline there are no imagenet / VGGFace weights: random He-normal stand-ins are used and can be
replaced through set_weights() with the keras `get_weights()` list of the sliced model."""
import numpy as np
import torch

from .. import functional as F
from ..nn import Net, he_normal
from ..ops import ACT_RELU, ConvSpec

VGG19_CFG = [64, 64, "P", 128, 128, "P", 256, 256, 256, 256, "P", 512, 512]
VGG16_CFG = [64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512]
VGG19_TAPS = (0, 1, 5, 9)    # conv ordinals of keras layers[1, 2, 8, 13]
VGG16_TAPS = (0, 1, 5, 8)    # conv ordinals of keras layers[1, 2, 8, 12]
C3 = ConvSpec((3, 3))


class VGGFeatures(Net):
    def __init__(self, cfg, taps, rng=None):
        super().__init__()
        rng = rng or np.random.default_rng(1234)
        self.cfg, self.taps = cfg, taps
        cin = 3
        for i, item in enumerate(c for c in cfg if c != "P"):
            self.add_weight("conv%d/kernel" % i, he_normal(rng, (3, 3, cin, item)), trainable=False)
            self.add_weight("conv%d/bias" % i, np.zeros(item, np.float32), trainable=False)
            cin = item
        self.finalize()

    def __call__(self, x_pre):
        feats, ci, x = [], 0, x_pre
        for item in self.cfg:
            if item == "P":
                x = F.maxpool(x, 2, 2)
            else:
                x = F.conv(x, self.weights[2 * ci], self.weights[2 * ci + 1], C3, ACT_RELU)
                if ci in self.taps:
                    feats.append(x)
                ci += 1
        return feats


class VGGLossFn(torch.autograd.Function):
    """sum over the tapped layers of mean((phi(x) - target)^2) (perceptual_loss.py:43-82) as ONE tape node (round 6): forward =
    the convolutions with bias + ReLU in their epilogues, the pools and one squared-difference reduction per tap; backward walks
    the frozen stack with ONE elementwise pass per layer -- at a tap  (g_next + 2 c (y - target)) relu'(y)  in a single kernel
    (cn_tap_bwd) instead of the loss term's gradient pass, autograd's add of the two gradients and the ReLU backward (9 -> 4
    tensor passes on the 134 MB taps of the benchmark's size).  `sizes` = None: one scalar over the whole batch; else the
    consecutive sample groups of PerceptualLoss.loss_groups, one scalar each.  Gradient w.r.t. x only (the stack is frozen).
    First-order only.  Inputs: net, x (preprocessed image), sizes, then the targets of the taps."""

    @staticmethod
    def forward(ctx, net, x, sizes, *targets):
        from .. import ops
        x = x.contiguous()
        ys, feats, ci, t = [], [], 0, x
        for item in net.cfg:
            if item == "P":
                t = ops.maxpool_fwd(t, 2, 2, 0)           # (its input is the previous convolution's output: ys[ci - 1])
            else:
                w, b = net.weights[2 * ci], net.weights[2 * ci + 1]
                t = ops.conv_fwd(t, w, b, C3.geom(tuple(t.shape), w.shape[-1]), ACT_RELU)
                ys.append(t)
                if ci in net.taps:
                    feats.append(t)
                ci += 1
        n = x.shape[0]
        groups = [n] if sizes is None else list(sizes)
        terms = []
        for a, b in zip(feats, targets):
            per = a.numel() // n
            n0, row = 0, []
            for sz in groups:
                row.append(ops.sqdiff_sum(a[n0:n0 + sz], b[n0:n0 + sz], 1.0 / (sz * per)))
                n0 += sz
            terms.append(torch.cat(row) if len(row) > 1 else row[0])
        total = torch.stack(terms).sum(0)
        ctx.net, ctx.sizes, ctx.x_shape = net, sizes, tuple(x.shape)
        ctx.save_for_backward(x, *ys, *targets)
        ctx.counts = (len(ys),)
        return total.reshape(()) if sizes is None else total

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        if torch.is_grad_enabled():
            raise RuntimeError("VGGLossFn is first-order only")
        net, sizes = ctx.net, ctx.sizes
        saved = ctx.saved_tensors
        (ny,) = ctx.counts
        x, ys, targets = saved[0], saved[1:1 + ny], saved[1 + ny:]
        n = x.shape[0]
        g = g.reshape(-1).float()
        tap_of = {ci: k for k, ci in enumerate(net.taps)}
        gcur, ci, pooled = None, ny, None
        for item in reversed(net.cfg):
            if item == "P":
                pooled, gcur = gcur, None             # the pool's backward joins the ReLU backward of the layer in front of it
                continue
            ci -= 1
            y = ys[ci]
            s_ = k = None
            if ci in tap_of:
                per = y.numel() // n
                if sizes is None:
                    s_, k = g.contiguous(), 2.0 / (n * per)          # one row = the whole tensor
                else:
                    s_ = torch.cat([g[i:i + 1].expand(sz) * (1.0 / (sz * per)) for i, sz in enumerate(sizes)]).contiguous()
                    k = 2.0
            gu = None
            if pooled is not None:
                # ReLU -> MaxPooling2D backward in ONE pass (the tap's own term added there too): no full-size gradient in between
                gu = ops.maxpool2_bwd_relu(y, pooled, targets[tap_of[ci]] if s_ is not None else None, s_, k if s_ is not None else 0.0)
                if gu is None:
                    gcur = ops.maxpool_bwd(y, pooled, 2, 2, 0)
                pooled = None
            if gu is not None:
                pass
            elif s_ is not None:
                gu = ops.tap_bwd(y, targets[tap_of[ci]], gcur, s_, k, ACT_RELU)
            elif gcur is None:
                continue                                             # (layers behind the last tap carry no gradient)
            else:
                gu = ops.act_bwd(gcur, y, ACT_RELU)
            w = net.weights[2 * ci]
            in_shape = conv_input_shape(net.cfg, ci, ctx.x_shape)
            gcur = ops.conv_dgrad(gu, w, C3.geom(in_shape, w.shape[-1]))
        return None, gcur, None, *([None] * len(targets))


def conv_input_shape(cfg, ci, x_shape):
    """(n, h, w, c) of the input of conv ordinal ci for a stack input of shape x_shape."""
    n, h, w, c = x_shape
    k = -1
    for item in cfg:
        if item == "P":
            h, w = h // 2, w // 2
        else:
            k += 1
            if k == ci:
                return (n, h, w, c)
            c = item
    raise IndexError(ci)
