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
