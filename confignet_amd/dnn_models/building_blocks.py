"""Building blocks (reference: confignet/dnn_models/building_blocks.py) on HIP kernels."""
import numpy as np
import torch

from .. import functional as F
from .. import ops
from ..nn import Net, glorot_uniform
from ..ops import ACT_LRELU, ACT_NONE, ConvSpec

KERAS_LRELU = 0.3      # keras.layers.LeakyReLU() default alpha (R1)
TF_LRELU = 0.2         # tf.nn.leaky_relu default / AdaIN MLP alpha (hologan_generator.py:21,56)


def mlp_weight_shapes(num_layers, num_in, num_hidden, num_out):
    shp, cur = [], num_in
    for _ in range(num_layers - 1):
        shp += [(cur, num_hidden), (num_hidden,)]
        cur = num_hidden
    return shp + [(cur, num_out), (num_out,)]


def mlp_forward(x, weights, alpha, fused=True):
    """MLPSimple.call (building_blocks.py:152-173): (Dense -> LeakyReLU(alpha)) x (L-1), Dense.
    fused=False keeps activation separate so the result is twice differentiable (R1 penalty)."""
    n_layers = len(weights) // 2
    for i in range(n_layers):
        w, b = weights[2 * i], weights[2 * i + 1]
        last = i == n_layers - 1
        if last:
            x = F.linear(x, w, b)
        elif fused:
            x = F.linear(x, w, b, ACT_LRELU, alpha)
        else:
            x = F.lrelu(F.linear(x, w, b), alpha)
    return x


class MLPSimple(Net):
    """Standalone MLP (latent discriminator, LatentGAN nets, per-input synthetic-encoder MLPs)."""

    def __init__(self, num_layers, num_in, num_hidden, num_out, alpha=KERAS_LRELU, rng=None):
        super().__init__()
        rng = rng or np.random.default_rng()
        self.num_in, self.num_out, self.alpha = num_in, num_out, alpha
        for i, s in enumerate(mlp_weight_shapes(num_layers, num_in, num_hidden, num_out)):
            self.add_weight("w%d" % i, glorot_uniform(rng, s) if len(s) == 2 else np.zeros(s, np.float32))
        self.finalize()

    def __call__(self, x, twice_differentiable=False):
        return mlp_forward(self.to_device(x), self.weights, self.alpha, fused=not twice_differentiable)

    def predict(self, x, batch_size=32):
        with torch.no_grad():
            return self(np.asarray(x, dtype=np.float32)).cpu().numpy()


def conv_adain(x, z, w6, spec, sb=None):
    """Conv3dAdaIn / Conv2dAdaIn.call (building_blocks.py:37-44,73-80): conv(same)+bias ->
    LeakyReLU(0.3) [fused epilogue] -> AdaIn with [s|b] = MLP(z) (LeakyReLU 0.2).  sb: the MLP's output when the caller has
    computed it already (the generator runs the MLPs of all its layers as one bank, F.mlp_bank)."""
    ck, cb, m0, b0, m1, b1 = w6
    if sb is None:
        sb = F.linear(F.linear(z, m0, b0, ACT_LRELU, TF_LRELU), m1, b1)
    with ops.request_stats("act"):                   # AdaIn's statistics from the convolution's epilogue where the launch carries them
        x = F.conv(x, ck, cb, spec, ACT_LRELU, KERAS_LRELU)
    return F.adain(x, sb)


DISCR_CONV = ConvSpec((3, 3), stride=2)


def discr_block(x, w4, return_styles, twice_differentiable=False, intermediates=None):
    """DiscrBlock.call (building_blocks.py:97-111): conv k3 s2 same; styles from the pre-activation
    output; LeakyReLU(0.3) then instance norm.  The fused tail is first-order only; with
    twice_differentiable=True every op is twice differentiable (composite path, kept as the cross-check of
    the tangent-pass R1).  `intermediates` (a list) receives the primal tensors the tangent pass reuses."""
    ck, cb, gamma, beta = w4
    in_shape = tuple(x.shape)
    if twice_differentiable:
        x = F.conv(x, ck, cb, DISCR_CONV)
    else:
        with ops.request_stats("pre4", KERAS_LRELU):     # the tail's four sums from the convolution's epilogue where the launch carries them
            x = F.conv(x, ck, cb, DISCR_CONV)
    if not twice_differentiable:
        y, style, mean, q, smean, ssd = F.DiscrTailFn.apply(x, gamma, beta, return_styles, KERAS_LRELU)
        if intermediates is not None:
            intermediates.append({"x": x, "mean": mean, "q": q, "smean": smean, "ssd": ssd, "in_shape": in_shape})
        return y, style
    styles = F.layer_style(x) if return_styles else None
    x = F.instance_norm(F.lrelu(x, KERAS_LRELU), gamma, beta)
    return x, styles
