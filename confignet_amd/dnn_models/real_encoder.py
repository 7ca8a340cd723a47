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

    def _bn_coefficients(self):
        """Per-channel affine of every conv+BN pair, computed for all 53 pairs at once (3 concatenations and
        5 elementwise launches on ~53k channels instead of ~5 launches per pair):
        bn(conv + b) = a*conv + shift,  a = gamma*rsqrt(var+eps),  shift = beta + a*(b - mean).
        The (C,) coefficient algebra is host-side plumbing and carries the gradients of gamma, beta and b."""
        ws = self.weights
        firsts = [f for f, _ in self._convs]
        sizes = [ws[f + 1].shape[0] for f in firsts]
        bias = torch.cat([ws[f + 1] for f in firsts])
        gamma = torch.cat([ws[f + 2] for f in firsts])
        beta = torch.cat([ws[f + 3] for f in firsts])
        if getattr(self, "_stat_cache", None) is None:
            self._stat_cache = (torch.cat([ws[f + 4] for f in firsts]), torch.cat([ws[f + 5] for f in firsts]))
        mean, var = self._stat_cache
        a = gamma * torch.rsqrt(var + BN_EPS)
        shift = beta + a * (bias - mean)
        return torch.split(a, sizes), torch.split(shift, sizes)

    def set_weights(self, weights):
        super().set_weights(weights)
        self._stat_cache = None

    def _conv_bn(self, ci, x, coef, res=None, relu=True):
        first, spec = self._convs[ci]
        z = F.conv(x, self.weights[first], None, spec)           # bias folded into the affine shift
        return F.channel_affine_act(z, coef[0][ci], coef[1][ci], res, relu)

    def features(self, img):
        coef = self._bn_coefficients()
        x = F.caffe_preprocess(img)                              # real_encoder.py:24-25
        x = self._conv_bn(0, x, coef)
        x = F.maxpool(x, 3, 2, 1)                                # pool1_pad + pool1_pool
        ci = 1
        for filters, blocks, stride1 in RESNET50_STACKS:
            for bi in range(blocks):
                if bi == 0:
                    sc = self._conv_bn(ci, x, coef, relu=False)
                    ci += 1
                else:
                    sc = x
                y = self._conv_bn(ci, x, coef)
                y = self._conv_bn(ci + 1, y, coef)
                x = self._conv_bn(ci + 2, y, coef, res=sc, relu=True)   # bn + add + relu in one pass
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
