"""Oracle networks: functional restatement of confignet/dnn_models/* on Keras-ordered
weight lists (Model.get_weights() order, R11).  TEST INFRASTRUCTURE (oracle/__init__.py).

Every function takes `w`, a list of torch tensors in the order the reference's
`get_weights()` would return them, and consumes it front to back.
"""
import numpy as np
import torch

from . import ref_ops as O


class _W:
    """Cursor over a Keras-ordered weight list."""

    def __init__(self, weights):
        self.w, self.i = weights, 0

    def take(self, n):
        out = self.w[self.i:self.i + n]
        assert len(out) == n, "weight list too short"
        self.i += n
        return out

    def done(self):
        assert self.i == len(self.w), "unused weights: %d of %d" % (self.i, len(self.w))


# ----------------------------------------------------------------------------
# HologanGenerator (hologan_generator.py:12-174)
# ----------------------------------------------------------------------------
def generator_forward(w, z, rotation, res, return_intermediates=False):
    """z: (N, L) or list of 5 (N, L); rotation (N, 3); res in {128, 256, 512}."""
    if isinstance(z, (list, tuple)):
        z30, z31, z20, z21, z22 = z
    else:
        z30 = z31 = z20 = z21 = z22 = z
    c = _W(w)
    n = z30.shape[0]
    inter = {}
    # learned_input: Dense(32768, kernel zeros-init, bias ones-init) on zeros(N,1) (l.24-27,133-136)
    k, b = c.take(2)
    x = O.stored(O.dense(torch.zeros(n, 1, dtype=z30.dtype), k, b).reshape(n, 4, 4, 4, 512))
    x = O.upsample2(x)                                                   # l.139
    # map_3d_0 / map_3d_1: Conv3D k3 same -> LeakyReLU(0.3) -> AdaIn (building_blocks.py:37-44)
    ck, cb, *mlp = c.take(6)
    x = O.stored(O.adain(O.stored(O.leaky_relu(O.conv_same(x, ck, cb), 0.3)), z30, mlp, 0.2))     # (O.stored: identity outside O.bf16_storage)
    inter["a3d0"] = x
    x = O.upsample2(x)                                                   # l.143
    ck, cb, *mlp = c.take(6)
    x = O.stored(O.adain(O.stored(O.leaky_relu(O.conv_same(x, ck, cb), 0.3)), z31, mlp, 0.2))
    inter["a3d1"] = x
    # rotation (l.147-148)
    x = O.stored(O.transform_3d_grid(x, O.euler_angles_to_matrix(rotation)))
    inter["rot"] = x
    # map_3d_post: 2x (Conv3D k3 + LeakyReLU(0.3)) (l.49-54,151)
    for _ in range(2):
        ck, cb = c.take(2)
        x = O.stored(O.leaky_relu(O.conv_same(x, ck, cb), 0.3))
    inter["post3d"] = x
    # depth collapse (l.153-156): (N,16,16,16,64) -> (N,16,16,1024), channel = d*64+c
    s = x.shape
    x = x.reshape(s[0], s[1], s[2], s[3] * s[4])
    # projection_conv 1x1 + tf.nn.leaky_relu (alpha 0.2) (l.56,157)
    ck, cb = c.take(2)
    x = O.stored(O.leaky_relu(O.conv_same(x, ck, cb), 0.2))
    inter["proj"] = x
    zs = [z20, z21, z22]
    if res > 128:
        zs.append(z22)
    if res > 256:
        zs.append(z22)
    for i, zz in enumerate(zs):                                          # l.159-170
        ck, cb, *mlp = c.take(6)
        x = O.stored(O.leaky_relu(O.conv_same(x, ck, cb), 0.3))
        x = O.stored(O.adain(x, zz, mlp, 0.2))
        inter["a2d%d" % i] = x
        x = O.upsample2(x)
    ck, cb = c.take(2)
    x = torch.tanh(O.conv_same(x, ck, cb))                               # map_final (l.101,172)
    c.done()
    if return_intermediates:
        return x, inter
    return x


def generator_weight_shapes(latent_dim, res, n_mlp_units=128):
    """Shapes in get_weights() order (attribute-assignment order of __init__)."""
    shp = [(1, 32768), (32768,)]

    def adain_mlp(c):
        return [(latent_dim, n_mlp_units), (n_mlp_units,), (n_mlp_units, 2 * c), (2 * c,)]

    shp += [(3, 3, 3, 512, 256), (256,)] + adain_mlp(256)
    shp += [(3, 3, 3, 256, 128), (128,)] + adain_mlp(128)
    shp += [(3, 3, 3, 128, 64), (64,), (3, 3, 3, 64, 64), (64,)]
    shp += [(1, 1, 1024, 512), (512,)]
    shp += [(4, 4, 512, 256), (256,)] + adain_mlp(256)
    shp += [(4, 4, 256, 64), (64,)] + adain_mlp(64)
    shp += [(4, 4, 64, 32), (32,)] + adain_mlp(32)
    last = 32
    if res > 128:
        shp += [(4, 4, 32, 32), (32,)] + adain_mlp(32)
    if res > 256:
        shp += [(4, 4, 32, 16), (16,)] + adain_mlp(16)
        last = 16
    shp += [(4, 4, last, 3), (3,)]
    return shp


# ----------------------------------------------------------------------------
# HologanDiscriminator / HologanLatentRegressor (hologan_discriminator.py)
# ----------------------------------------------------------------------------
def discr_channels(n_layers=5, f0=48, fmax=512):
    return [min(f0 * 2 ** i, fmax) for i in range(n_layers)]


def discr_block(x, ck, cb, gamma, beta, return_styles):
    """DiscrBlock (building_blocks.py:83-111): conv k3 s2 same; styles from the
    pre-activation output; LeakyReLU(0.3) THEN instance norm."""
    x = O.stored(O.conv_same(x, ck, cb, stride=2))
    styles = None
    if return_styles:
        mu, std = O.layer_style(x)
        styles = torch.cat([mu, std], dim=-1).reshape(x.shape[0], -1)    # (N, 2C): [mu | std]
    x = O.leaky_relu(x, 0.3)
    x = O.stored(O.instance_norm(x, gamma, beta))
    return x, styles


def discriminator_forward(w, img, n_layers=5):
    """Returns an insertion-ordered dict discr_style_0..4, discr_final (l.48-64)."""
    c = _W(w)
    ck, cb = c.take(2)
    x = O.conv_same(img, ck, cb)                                         # initial 1x1 conv (l.20,50)
    blocks = [c.take(4) for _ in range(n_layers)]
    heads = [c.take(2) for _ in range(n_layers)]
    fk, fb = c.take(2)
    c.done()
    out = {}
    for i in range(n_layers):
        x, st = discr_block(x, *blocks[i], return_styles=True)
        out["discr_style_%d" % i] = O.dense(st, *heads[i])
    x = x.reshape(x.shape[0], -1)                                        # (H, W, C) flatten order
    out["discr_final"] = O.dense(x, fk, fb)
    return out


def discriminator_weight_shapes(res, n_layers=5, f0=48, fmax=512):
    ch = discr_channels(n_layers, f0, fmax)
    shp = [(1, 1, 3, 3), (3,)]
    cin = 3
    for c in ch:
        shp += [(3, 3, cin, c), (c,), (c,), (c,)]
        cin = c
    for c in ch:
        shp += [(2 * c, 1), (1,)]
    out = res // 2 ** n_layers
    shp += [(min(2 ** n_layers * fmax // 2, fmax) * out * out, 1), (1,)]
    return shp


def latent_regressor_forward(w, img, n_layers=5):
    c = _W(w)
    ck, cb = c.take(2)
    x = O.conv_same(img, ck, cb)
    for _ in range(n_layers):
        x, _ = discr_block(x, *c.take(4), return_styles=False)
    fk, fb = c.take(2)
    c.done()
    return O.dense(x.reshape(x.shape[0], -1), fk, fb)


def latent_regressor_weight_shapes(latent_dim, res, n_layers=5, f0=48, fmax=512):
    ch = discr_channels(n_layers, f0, fmax)
    shp = [(1, 1, 3, 3), (3,)]
    cin = 3
    for c in ch:
        shp += [(3, 3, cin, c), (c,), (c,), (c,)]
        cin = c
    out = res // 2 ** n_layers
    shp += [(min(2 ** n_layers * fmax // 2, fmax) * out * out, latent_dim + 3), (latent_dim + 3,)]
    return shp


# ----------------------------------------------------------------------------
# SyntheticDataEncoder (synthetic_encoder.py) / latent discriminator
# ----------------------------------------------------------------------------
def synthetic_encoder_forward(w, params):
    """params: list of (N, in_i) in facemodel_param_names (sorted) order; each MLP is
    2 layers in->in (LeakyReLU 0.3)->out; outputs concatenated along axis 1."""
    c = _W(w)
    outs = [O.mlp_simple(p, c.take(4), 0.3) for p in params]
    c.done()
    return torch.cat(outs, dim=1)


def synthetic_encoder_weight_shapes(io_dims):
    shp = []
    for din, dout in io_dims:
        shp += [(din, din), (din,), (din, dout), (dout,)]
    return shp


def mlp_weight_shapes(n_layers, n_in, n_hidden, n_out):
    shp, cur = [], n_in
    for _ in range(n_layers - 1):
        shp += [(cur, n_hidden), (n_hidden,)]
        cur = n_hidden
    shp += [(cur, n_out), (n_out,)]
    return shp


# ----------------------------------------------------------------------------
# keras.applications VGG19 / VGG16 slices (perceptual_loss.py:18-41)  [TF-2.1]
# ----------------------------------------------------------------------------
VGG19_CFG = [64, 64, "P", 128, 128, "P", 256, 256, 256, 256, "P", 512, 512]   # to block4_conv2
VGG16_CFG = [64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512]        # to block4_conv2
VGG19_TAPS = [0, 1, 5, 9]     # conv indices of layers[1,2,8,13]: b1c1, b1c2, b3c2, b4c2
VGG16_TAPS = [0, 1, 5, 8]     # conv indices of layers[1,2,8,12]: b1c1, b1c2, b3c2, b4c2


def vgg_features(w, x_pre, cfg, taps):
    """x_pre: preprocessed image (N,H,W,3).  Returns the tapped post-ReLU activations."""
    c = _W(w)
    feats, ci, x = [], 0, x_pre
    for item in cfg:
        if item == "P":
            x = O.maxpool(x, 2, 2)
        else:
            ck, cb = c.take(2)
            x = O.relu(O.conv_same(x, ck, cb))
            if ci in taps:
                feats.append(x)
            ci += 1
    c.done()
    return feats


def vgg_weight_shapes(cfg):
    shp, cin = [], 3
    for item in cfg:
        if item != "P":
            shp += [(3, 3, cin, item), (item,)]
            cin = item
    return shp


def perceptual_loss(w, a, b, model_type="imagenet"):
    """PerceptualLoss.loss (perceptual_loss.py:43-82): sum over taps of the global mean
    squared difference of activations."""
    if model_type == "imagenet":
        pre, cfg, taps = O.caffe_preprocess, VGG19_CFG, VGG19_TAPS
    else:
        pre, cfg, taps = O.vggface_preprocess, VGG16_CFG, VGG16_TAPS
    fa = vgg_features(w, pre(a), cfg, taps)
    fb = vgg_features(w, pre(b), cfg, taps)
    total = 0
    for x, y in zip(fa, fb):
        total = total + ((x - y) ** 2).mean()
    return total


# ----------------------------------------------------------------------------
# keras.applications ResNet50 v1 (resnet_common) + RealEncoder (real_encoder.py)  [TF-2.1]
# ----------------------------------------------------------------------------
RESNET50_STACKS = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]
BN_EPS = 1.001e-5


def _conv_bn(conv, bn, x, stride=1, relu=True, pad7=False):
    ck, cb = conv
    g, b, m, v = bn
    if pad7:
        x = O.conv_valid_padded(x, ck, cb, stride, 3)
    else:
        x = O.conv_same(x, ck, cb, stride=stride)
    x = O.bn_inference(x, g, b, m, v, BN_EPS)
    return O.relu(x) if relu else x


def resnet50_layer_order():
    """Weight-carrying layers of keras.applications ResNet50 (resnet_common.ResNet50, include_top=False) in
    `model.layers` order [TF-2.1]: a functional model sorts its layers by depth from the output and, inside one depth,
    by the order of a depth-first walk from the output that visits Add's inputs shortcut first (network.py:
    _map_graph_network).  In a conv-shortcut block that gives
        <b>_1_conv, <b>_1_bn, <b>_2_conv, <b>_2_bn, <b>_0_conv, <b>_3_conv, <b>_0_bn, <b>_3_bn
    (what `ResNet50().summary()` prints), NOT shortcut first.  Returns [(name, kind, cin, cout, k)]."""
    order = [("conv1_conv", "conv", 3, 64, 7), ("conv1_bn", "bn", 0, 64, 0)]
    cin = 64
    for si, (filters, blocks, _) in enumerate(RESNET50_STACKS):
        for bi in range(blocks):
            n = "conv%d_block%d" % (si + 2, bi + 1)
            f = filters
            order += [(n + "_1_conv", "conv", cin, f, 1), (n + "_1_bn", "bn", 0, f, 0),
                      (n + "_2_conv", "conv", f, f, 3), (n + "_2_bn", "bn", 0, f, 0)]
            if bi == 0:
                order += [(n + "_0_conv", "conv", cin, 4 * f, 1), (n + "_3_conv", "conv", f, 4 * f, 1),
                          (n + "_0_bn", "bn", 0, 4 * f, 0), (n + "_3_bn", "bn", 0, 4 * f, 0)]
            else:
                order += [(n + "_3_conv", "conv", f, 4 * f, 1), (n + "_3_bn", "bn", 0, 4 * f, 0)]
            cin = 4 * f
    return order


def resnet50_features(w_cursor, x_pre):
    c = w_cursor
    L = {}
    for name, kind, _, _, _ in resnet50_layer_order():      # Conv2D: kernel, bias; BatchNormalization: gamma, beta, mean, var
        L[name] = c.take(2 if kind == "conv" else 4)

    def cb(prefix, x, **kw):
        return _conv_bn(L[prefix + "_conv"], L[prefix + "_bn"], x, **kw)

    x = cb("conv1", x_pre, stride=2, pad7=True)          # conv1_pad(3) + 7x7 s2 valid + bn + relu
    x = O.maxpool(x, 3, 2, pad=1)                        # pool1_pad(1) + 3x3 s2
    for si, (filters, blocks, stride1) in enumerate(RESNET50_STACKS):
        for bi in range(blocks):
            n = "conv%d_block%d" % (si + 2, bi + 1)
            s = stride1 if bi == 0 else 1
            sc = cb(n + "_0", x, stride=s, relu=False) if bi == 0 else x     # 0_conv 1x1 (4f, stride) + 0_bn
            y = cb(n + "_1", x, stride=s)                                    # 1_conv 1x1 (f, stride)
            y = cb(n + "_2", y)                                              # 2_conv 3x3 same
            y = cb(n + "_3", y, relu=False)                                  # 3_conv 1x1 (4f)
            x = O.relu(sc + y)
    return x.mean(dim=(1, 2))                                                # pooling="avg"


def resnet50_weight_shapes():
    shp = []
    for _, kind, cin, cout, k in resnet50_layer_order():
        shp += [(k, k, cin, cout), (cout,)] if kind == "conv" else [(cout,)] * 4
    return shp


def resnet50_weight_roles():
    """Per entry of resnet50_weight_shapes(): 'kernel' | 'bias' | 'gamma' | 'beta' | 'mean' | 'var'."""
    roles = []
    for _, kind, _, _, _ in resnet50_layer_order():
        roles += ["kernel", "bias"] if kind == "conv" else ["gamma", "beta", "mean", "var"]
    return roles


def real_encoder_forward(w, img, rotation_ranges=((-30, 30), (-10, 10), (0, 0))):
    """RealEncoder.call (real_encoder.py:23-34). Returns (embedding (N,L), rotation (N,3))."""
    c = _W(w)
    feat = resnet50_features(c, O.caffe_preprocess(img))
    rk, rb, lk, lb = c.take(4)
    c.done()
    mult = np.pi * np.array([r[1] for r in rotation_ranges]) / 180.0
    rot = torch.tanh(O.dense(feat, rk, rb)) * torch.tensor(mult, dtype=img.dtype)
    return O.dense(feat, lk, lb), rot


def real_encoder_weight_shapes(latent_dim):
    return resnet50_weight_shapes() + [(2048, 3), (3,), (2048, latent_dim), (latent_dim,)]
