"""Second, independent restatement of whole NETWORKS in explicit NumPy float64, built only from oracle/np_ops.py (tap loops,
per-voxel resampling) and written from the reference files, not from oracle/ref_nets.py: HologanGenerator.call
(confignet/dnn_models/hologan_generator.py:129-174), HologanDiscriminator.call (hologan_discriminator.py:48-64), DiscrBlock
(building_blocks.py:83-111), MLPSimple / AdaIn (building_blocks.py:114-173).  No autograd exists here: derivatives are taken by
central finite differences of these forward passes (tests/test_oracle_kat.py), which is what pins the torch oracle's backward --
including r1_penalty's gradient-of-gradient and the rotation gradient through the interpolation weights -- independently of
torch.autograd.  TEST INFRASTRUCTURE (oracle/__init__.py); seconds per call, tiny shapes only."""
import numpy as np

from . import np_ops as NP


class _Cursor:
    def __init__(self, weights):
        self.w, self.i = [np.asarray(a, np.float64) for a in weights], 0

    def take(self, n):
        out = self.w[self.i:self.i + n]
        self.i += n
        return out

    def done(self):
        assert self.i == len(self.w), "weight list length mismatch: used %d of %d" % (self.i, len(self.w))


def mlp(x, weights, alpha):
    """MLPSimple.call (building_blocks.py:166-173): Dense -> LeakyReLU(alpha) for every layer but the last, then Dense."""
    n = len(weights) // 2
    for i in range(n):
        x = x @ weights[2 * i] + weights[2 * i + 1]
        if i < n - 1:
            x = NP.leaky_relu(x, alpha)
    return x


def adain(x, z, mlp_weights):
    """AdaIn.call (building_blocks.py:135-149): LayerNormalization over the spatial axes without affine parameters (epsilon
    1e-3 inside the square root, tf.keras default), scale / bias = the two halves of MLP(z) (MLP alpha 0.2), x * (scale + 1) + bias."""
    c = x.shape[-1]
    sb = mlp(z, mlp_weights, 0.2)
    shape = (x.shape[0],) + (1,) * (x.ndim - 2) + (c,)
    return NP.layer_norm_spatial(x, 1e-3) * (sb[:, :c].reshape(shape) + 1.0) + sb[:, c:].reshape(shape)


def generator_forward(weights, z, rotation, res):
    """hologan_generator.py:129-174 with one latent vector for all five AdaIn inputs (z: (N, L)), rotation (N, 3) Euler angles."""
    z, rotation = np.asarray(z, np.float64), np.asarray(rotation, np.float64)
    c = _Cursor(weights)
    n = z.shape[0]
    k, b = c.take(2)                                                  # learned_input: Dense on zeros(N, 1) -> its bias (l.133-136)
    x = (np.zeros((n, 1)) @ k + b).reshape(n, 4, 4, 4, 512)
    for _ in range(2):                                                # UpSampling3D -> Conv3dAdaIn (l.139-144)
        x = NP.upsample2(x)
        ck, cb, *m = c.take(6)
        x = adain(NP.leaky_relu(NP.conv_same(x, ck, cb), 0.3), z, m)  # building_blocks.py:37-44: conv, LeakyReLU() = 0.3, AdaIn
    x = NP.transform_3d_grid(x, NP.euler_angles_to_matrix(rotation))  # l.147-148
    for _ in range(2):                                                # map_3d_post (l.49-54, 151)
        ck, cb = c.take(2)
        x = NP.leaky_relu(NP.conv_same(x, ck, cb), 0.3)
    x = x.reshape(n, x.shape[1], x.shape[2], x.shape[3] * x.shape[4])  # l.153-156: depth folded into channels
    ck, cb = c.take(2)
    x = NP.leaky_relu(NP.conv_same(x, ck, cb), 0.2)                   # projection conv + tf.nn.leaky_relu (l.56, 157)
    n_2d = 3 + (res > 128) + (res > 256)
    for _ in range(n_2d):                                             # Conv2dAdaIn, then UpSampling2D (l.159-170)
        ck, cb, *m = c.take(6)
        x = adain(NP.leaky_relu(NP.conv_same(x, ck, cb), 0.3), z, m)
        x = NP.upsample2(x)
    ck, cb = c.take(2)
    x = np.tanh(NP.conv_same(x, ck, cb))                              # map_final (l.101, 172)
    c.done()
    return x


def discr_block(x, ck, cb, gamma, beta):
    """DiscrBlock.call (building_blocks.py:97-111): Conv2D k3 stride 2 same; the layer style (per-channel mean and sqrt(var +
    1e-6) over space, confignet_utils.get_layer_style) of the PRE-activation tensor as (N, 2C) = [mean | std]; LeakyReLU() = 0.3;
    InstanceNormalization with epsilon 1e-3 added to the standard deviation (instance_normalization.py:108-131)."""
    x = NP.conv_same(x, ck, cb, stride=2)
    mu, sd = NP.layer_style(x)
    styles = np.concatenate([mu.reshape(x.shape[0], -1), sd.reshape(x.shape[0], -1)], axis=1)
    x = NP.instance_norm(NP.leaky_relu(x, 0.3), gamma, beta, 1e-3)
    return x, styles


def discriminator_forward(weights, img, n_layers=5):
    """hologan_discriminator.py:48-64: 1x1 from-RGB conv, n_layers DiscrBlocks with a Dense(1) head on each block's styles, a
    Dense(1) head on the flattened last activation.  Weight order (l.10-46): from-RGB, blocks, style heads, final head."""
    c = _Cursor(weights)
    ck, cb = c.take(2)
    x = NP.conv_same(np.asarray(img, np.float64), ck, cb)
    blocks = [c.take(4) for _ in range(n_layers)]
    heads = [c.take(2) for _ in range(n_layers)]
    fk, fb = c.take(2)
    c.done()
    out = []
    for i in range(n_layers):
        x, st = discr_block(x, *blocks[i])
        out.append(st @ heads[i][0] + heads[i][1])
    out.append(x.reshape(x.shape[0], -1) @ fk + fb)
    return out


def central_difference(f, x, idx, h):
    """d f / d x[idx] of a scalar function by the five-point stencil (truncation O(h^4))."""
    x = np.array(x, np.float64)
    old = x[idx]
    vals = []
    for k in (-2, -1, 1, 2):
        x[idx] = old + k * h
        vals.append(f(x))
    x[idx] = old
    return (vals[0] - 8 * vals[1] + 8 * vals[2] - vals[3]) / (12 * h)
