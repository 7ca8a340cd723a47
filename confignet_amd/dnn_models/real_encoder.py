"""RealEncoder (reference: confignet/dnn_models/real_encoder.py): keras.applications ResNet50 v1
(include_top=False, pooling="avg") + rotation / latent heads, on HIP kernels.

[TF-2.1] The subclassed model is called without `training=`, so BatchNormalization runs in
inference mode on its (never updated) moving statistics while gamma/beta and all conv kernels
are trained: BN folds into a per-channel affine applied after each conv."""
import numpy as np
import torch

from .. import functional as F
from ..nn import Net, glorot_uniform, he_normal
from ..ops import ACT_TANH, ConvSpec

RESNET50_STACKS = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]
BN_EPS = 1.001e-5
C7 = ConvSpec((7, 7), stride=2, explicit_pad=3)
C1 = {1: ConvSpec((1, 1)), 2: ConvSpec((1, 1), stride=2)}
C3 = ConvSpec((3, 3))


class RealEncoder(Net):
    def __init__(self, latent_dim, input_shape, rotation_ranges, rng=None):
        super().__init__()
        rng = rng or np.random.default_rng()
        self._convs = []           # (first weight index, spec)

        def conv_bn(k, cin, cout, spec, last_in_block=False):
            first = len(self._entries)
            self.add_weight("conv%d/kernel" % first, he_normal(rng, (k, k, cin, cout)))
            self.add_weight("conv%d/bias" % first, np.zeros(cout, np.float32))
            # a residual branch's last BN starts small so 16 stacked blocks keep O(1) activations
            self.add_weight("bn%d/gamma" % first, np.full(cout, 0.25 if last_in_block else 1.0, np.float32))
            self.add_weight("bn%d/beta" % first, np.zeros(cout, np.float32))
            self.add_weight("bn%d/moving_mean" % first, np.zeros(cout, np.float32), trainable=False)
            self.add_weight("bn%d/moving_variance" % first, np.ones(cout, np.float32), trainable=False)
            self._convs.append((first, spec))

        conv_bn(7, 3, 64, C7)
        cin = 64
        for filters, blocks, stride1 in RESNET50_STACKS:
            for bi in range(blocks):
                s = stride1 if bi == 0 else 1
                if bi == 0:
                    conv_bn(1, cin, 4 * filters, C1[s])          # 0_conv shortcut
                conv_bn(1, cin, filters, C1[s])                  # 1_conv (stride on the first 1x1)
                conv_bn(3, filters, filters, C3)                 # 2_conv
                conv_bn(1, filters, 4 * filters, C1[1], True)    # 3_conv
                cin = 4 * filters
        self.resnet_feature_dim = 2048
        self.add_weight("rotation_regressor/kernel", glorot_uniform(rng, (2048, 3)))
        self.add_weight("rotation_regressor/bias", np.zeros(3, np.float32))
        self.add_weight("feature_to_latent_mlp/kernel", glorot_uniform(rng, (2048, latent_dim)))
        self.add_weight("feature_to_latent_mlp/bias", np.zeros(latent_dim, np.float32))
        self.finalize()
        mult = np.pi * np.array([rotation_ranges[0][1], rotation_ranges[1][1], rotation_ranges[2][1]]) / 180.0
        self.rotation_range_multiplier = torch.tensor(mult, dtype=torch.float32, device=self.device)

    def _conv_bn(self, ci, x, res=None, relu=True):
        first, spec = self._convs[ci]
        k, b, gamma, beta, mean, var = self.weights[first:first + 6]
        # conv bias and BN (inference) fold into ONE per-channel affine after the bias-free conv:
        # bn(conv + b) = a*conv + (beta + a*(b - mean)); the (C,) coefficient algebra is host-side plumbing and
        # carries the gradients of gamma, beta and the conv bias (no activation-sized bias-gradient pass).
        z = F.conv(x, k, None, spec)
        a = gamma * torch.rsqrt(var + BN_EPS)
        return F.channel_affine_act(z, a, beta + a * (b - mean), res, relu)

    def features(self, img):
        x = F.caffe_preprocess(img)                              # real_encoder.py:24-25
        x = self._conv_bn(0, x)
        x = F.maxpool(x, 3, 2, 1)                                # pool1_pad + pool1_pool
        ci = 1
        for filters, blocks, stride1 in RESNET50_STACKS:
            for bi in range(blocks):
                if bi == 0:
                    sc = self._conv_bn(ci, x, relu=False)
                    ci += 1
                else:
                    sc = x
                y = self._conv_bn(ci, x)
                y = self._conv_bn(ci + 1, y)
                x = self._conv_bn(ci + 2, y, res=sc, relu=True)   # bn + add + relu in one pass
                ci += 3
        return F.global_avg_pool(x)

    def __call__(self, input_img):
        feat = self.features(self.to_device(input_img))
        w = self.weights
        rot = F.linear(feat, w[-4], w[-3], ACT_TANH) * self.rotation_range_multiplier
        return F.linear(feat, w[-2], w[-1]), rot

    def predict(self, imgs, batch_size=32):
        embs, rots = [], []
        with torch.no_grad():
            for s in range(0, len(imgs), batch_size):
                e, r = self(imgs[s:s + batch_size])
                embs.append(e.cpu().numpy())
                rots.append(r.cpu().numpy())
        return np.concatenate(embs), np.concatenate(rots)
