"""HologanDiscriminator / HologanLatentRegressor (reference:
confignet/dnn_models/hologan_discriminator.py) on HIP kernels."""
from collections import OrderedDict

import numpy as np
import torch

from .. import functional as F
from ..nn import Net, glorot_uniform
from ..ops import ConvSpec
from .building_blocks import discr_block

C1 = ConvSpec((1, 1))


def _trunk_weights(net, rng, num_resample, f0, fmax, ksize):
    chans, cin = [], 3
    net.add_weight("initial_1x1_conv/kernel", glorot_uniform(rng, (1, 1, 3, 3)))
    net.add_weight("initial_1x1_conv/bias", np.zeros(3, np.float32))
    e = 1
    for i in range(num_resample):
        c = int(min(e * f0, fmax))
        net.add_weight("block%d/kernel" % i, glorot_uniform(rng, (ksize, ksize, cin, c)))
        net.add_weight("block%d/bias" % i, np.zeros(c, np.float32))
        net.add_weight("block%d/gamma" % i, np.ones(c, np.float32))
        net.add_weight("block%d/beta" % i, np.zeros(c, np.float32))
        chans.append(c)
        cin, e = c, e * 2
    return chans, e


class HologanDiscriminator(Net):
    def __init__(self, img_shape, num_resample, disc_max_feature_maps, disc_kernel_size, disc_expansion_factor,
                 initial_from_rgb_layer_in_discr, rng=None):
        super().__init__()
        assert initial_from_rgb_layer_in_discr and disc_kernel_size == 3
        rng = rng or np.random.default_rng()
        self.num_resample = num_resample
        self.out_size = (int(img_shape[0] / 2 ** num_resample), int(img_shape[1] / 2 ** num_resample))
        chans, e = _trunk_weights(self, rng, num_resample, disc_expansion_factor, disc_max_feature_maps, disc_kernel_size)
        for i, c in enumerate(chans):
            self.add_weight("style_classifier%d/kernel" % i, glorot_uniform(rng, (2 * c, 1)))
            self.add_weight("style_classifier%d/bias" % i, np.zeros(1, np.float32))
        self.num_linear_in = int(min(e * disc_max_feature_maps // 2, disc_max_feature_maps)) * self.out_size[0] * self.out_size[1]
        self.add_weight("disc_map/kernel", glorot_uniform(rng, (self.num_linear_in, 1)))
        self.add_weight("disc_map/bias", np.zeros(1, np.float32))
        self.finalize()

    def __call__(self, input_img, twice_differentiable=False, intermediates=None):
        """Returns the insertion-ordered dict discr_style_0..n-1, discr_final (l.48-64)."""
        w = self.weights
        nr = self.num_resample
        x = F.conv(self.to_device(input_img), w[0], w[1], C1)
        heads = 2 + 4 * nr
        out = OrderedDict()
        for i in range(nr):
            x, st = discr_block(x, w[2 + 4 * i:6 + 4 * i], True, twice_differentiable, intermediates)
            out["discr_style_%d" % i] = F.linear(st, w[heads + 2 * i], w[heads + 2 * i + 1])
        x = x.reshape(x.shape[0], -1)
        out["discr_final"] = F.linear(x, w[-2], w[-1])
        return out

    def input_gradients(self, intermediates, stacked=False):
        """[d sum_n out_i / d image for every head i] -- what losses.py:75-82 asks the inner tape for, head by head
        (`tape.gradient(out_i, real_imgs)`) -- computed for ALL heads in ONE backward sweep without a tape: the cotangents of
        the heads still "above" a block are stacked along the batch axis (head b joins at block b through its style
        statistics), so every block runs one statistics pass, one affine pass and ONE data-gradient convolution on
        (6 - b) * N samples instead of 6 - b separate launches on N (the activations are read through a sample period, not
        copied).  Constants of the outer tape: the callers detach these gradients anyway (the penalty's weight gradient
        flows through `tangent`)."""
        from .. import ops
        from .building_blocks import DISCR_CONV, KERAS_LRELU
        w, nr = self.weights, self.num_resample
        heads = 2 + 4 * nr
        n = intermediates[0]["x"].shape[0]
        with torch.no_grad():
            last = intermediates[-1]["x"]
            # final head: out = flatten(y_last) @ W + b  ->  d out / d y_last = W, the same for every sample
            G = w[-2].detach().reshape(1, *last.shape[1:]).expand(n, *last.shape[1:]).contiguous()
            for b in reversed(range(nr)):
                it = intermediates[b]
                x = it["x"]
                sp, c = x.numel() // (x.shape[0] * x.shape[-1]), x.shape[-1]
                h = G.shape[0] // n                                   # heads above this block
                out = torch.empty(((h + 1) * n, *x.shape[1:]), device=x.device, dtype=x.dtype)
                # heads above: instance norm + LeakyReLU backward (building_blocks.py:105-106), all of them at once
                t1, t2 = ops.nc_reduce(G, x, flags=2, slope=KERAS_LRELU, x2_period=n)
                c1, c2, c0, _, _ = ops.norm_coef_bwd(ops.NORM_INSTANCE, t1, t2, it["mean"].repeat(h, 1), it["q"].repeat(h, 1),
                                                     w[4 + 4 * b].detach(), sp, 1e-3)
                ops.nc_lin2((h * n, *x.shape[1:]), G, c1, x, c2, c0, flags=2 | 4, slope=KERAS_LRELU, x2_period=n, out=out[n:])
                # head b enters through the style statistics of this block's pre-activation (building_blocks.py:100-102)
                gstyle = w[heads + 2 * b].detach().reshape(1, 2 * c).expand(n, 2 * c).contiguous()
                _, d2, d0, _, _ = ops.norm_coef_bwd(ops.NORM_STYLE, gstyle, None, it["smean"], it["ssd"], None, sp, 1e-6)
                ops.nc_lin2(tuple(x.shape), x, d2, b=d0, out=out[:n])
                # the block's stride-2 convolution: one data-gradient launch for every head
                k = w[2 + 4 * b]
                in_shape = (out.shape[0],) + tuple(it["in_shape"][1:])
                G = ops.conv_dgrad(out, k.detach(), DISCR_CONV.geom(in_shape, c))
            g_img = ops.conv_dgrad(G, w[0].detach(), C1.geom(tuple(G.shape), 3))     # from-RGB 1x1 convolution
        if stacked:
            return g_img                           # (6 N, H, W, 3), head-major: style heads 0..n-1, then the final head
        return [g_img[i * n:(i + 1) * n] for i in range(nr + 1)]

    def tangent_all(self, v, intermediates):
        """The JVPs of ALL heads in one tangent pass (round 3): v = the stacked input gradients (input_gradients(..., stacked=True):
        (heads * N, H, W, 3), head-major), each head's tangent pushed in ITS OWN direction.  At block k the stack holds heads
        k..n: head k leaves through the block's style statistics, the others go on -- one convolution per block on the stack
        (M grows up to 6x: the 1024..4096-row layers of the deep blocks fill the chip without split-K) and, in the backward pass,
        one data gradient and one filter gradient per block instead of one per head.  Returns [jvp_0 .. jvp_n], each (N, 1);
        same arithmetic as `tangent` head by head (kept as the cross-check)."""
        from .building_blocks import DISCR_CONV, KERAS_LRELU
        w, nr = self.weights, self.num_resample
        heads = 2 + 4 * nr
        n = intermediates[0]["x"].shape[0]
        assert v.shape[0] == (nr + 1) * n
        out = []
        t = F.conv(v, w[0], None, C1)
        for k in range(nr):
            tx = F.conv(t, w[2 + 4 * k], None, DISCR_CONV)
            it = intermediates[k]
            t, tstyle = F.DualTailBatchedFn.apply(tx, it["x"], w[4 + 4 * k], it["mean"], it["q"], it["smean"], it["ssd"], KERAS_LRELU)
            out.append(F.linear(tstyle, w[heads + 2 * k], None))
        out.append(F.linear(t.reshape(t.shape[0], -1), w[-2], None))
        return out

    def tangent(self, v, intermediates, head):
        """JVP of output `head` (0..n-1 style heads, n = final head) w.r.t. the input image in direction v,
        evaluated at the primal pass that filled `intermediates`.  Linear layers act on the tangent without
        bias; the DiscrBlock tail uses DualTailFn.  Returns (N, 1)."""
        from .building_blocks import DISCR_CONV, KERAS_LRELU
        w = self.weights
        nr = self.num_resample
        heads = 2 + 4 * nr
        t = F.conv(v, w[0], None, C1)
        for k in range(nr):
            tx = F.conv(t, w[2 + 4 * k], None, DISCR_CONV)
            it = intermediates[k]
            style_head = head == k
            ty, tstyle = F.DualTailFn.apply(tx, it["x"], w[4 + 4 * k], it["mean"], it["q"], it["smean"], it["ssd"],
                                            not style_head, style_head, KERAS_LRELU)
            if style_head:
                return F.linear(tstyle, w[heads + 2 * k], None)
            t = ty
        return F.linear(t.reshape(t.shape[0], -1), w[-2], None)

    def predict(self, x, batch_size=32):
        with torch.no_grad():
            return {k: v.cpu().numpy() for k, v in self(np.asarray(x, np.float32)).items()}


class HologanLatentRegressor(Net):
    def __init__(self, latent_dim, img_shape, num_resample, disc_max_feature_maps, disc_kernel_size,
                 disc_expansion_factor, initial_from_rgb_layer_in_discr, rng=None):
        super().__init__()
        assert initial_from_rgb_layer_in_discr and disc_kernel_size == 3
        rng = rng or np.random.default_rng()
        self.num_resample = num_resample
        self.out_size = (int(img_shape[0] / 2 ** num_resample), int(img_shape[1] / 2 ** num_resample))
        _, e = _trunk_weights(self, rng, num_resample, disc_expansion_factor, disc_max_feature_maps, disc_kernel_size)
        self.num_linear_in = int(min(e * disc_max_feature_maps // 2, disc_max_feature_maps)) * self.out_size[0] * self.out_size[1]
        self.add_weight("latent_predictor/kernel", glorot_uniform(rng, (self.num_linear_in, latent_dim + 3)))
        self.add_weight("latent_predictor/bias", np.zeros(latent_dim + 3, np.float32))
        self.finalize()

    def __call__(self, inputs):
        w = self.weights
        x = F.conv(self.to_device(inputs), w[0], w[1], C1)
        for i in range(self.num_resample):
            x, _ = discr_block(x, w[2 + 4 * i:6 + 4 * i], False)
        return F.linear(x.reshape(x.shape[0], -1), w[-2], w[-1])

    def predict(self, x, batch_size=32):
        with torch.no_grad():
            return self(np.asarray(x, np.float32)).cpu().numpy()
