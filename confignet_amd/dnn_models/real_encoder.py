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


def resnet50_layers():
    """Weight-carrying layers of keras.applications ResNet50 (include_top=False) in `model.layers` order, which is
    the order of `get_weights()` and therefore of the `real_encoder_weights` list in a checkpoint
    (confignet_second_stage.py:35-43).  [TF-2.1] A functional model sorts layers by depth from the output; inside one
    depth by a depth-first walk from the output that visits the Add's shortcut input first, so a conv-shortcut block
    lists  _1_conv, _1_bn, _2_conv, _2_bn, _0_conv, _3_conv, _0_bn, _3_bn  (as `ResNet50().summary()` prints it).
    Yields (name, kind, cin, cout, k)."""
    yield "conv1_conv", "conv", 3, 64, 7
    yield "conv1_bn", "bn", 0, 64, 0
    cin = 64
    for si, (f, blocks, _) in enumerate(RESNET50_STACKS):
        for bi in range(blocks):
            n = "conv%d_block%d" % (si + 2, bi + 1)
            yield n + "_1_conv", "conv", cin, f, 1
            yield n + "_1_bn", "bn", 0, f, 0
            yield n + "_2_conv", "conv", f, f, 3
            yield n + "_2_bn", "bn", 0, f, 0
            if bi == 0:
                yield n + "_0_conv", "conv", cin, 4 * f, 1
                yield n + "_3_conv", "conv", f, 4 * f, 1
                yield n + "_0_bn", "bn", 0, 4 * f, 0
                yield n + "_3_bn", "bn", 0, 4 * f, 0
            else:
                yield n + "_3_conv", "conv", f, 4 * f, 1
                yield n + "_3_bn", "bn", 0, 4 * f, 0
            cin = 4 * f


class RealEncoder(Net):
    def __init__(self, latent_dim, input_shape, rotation_ranges, rng=None):
        super().__init__()
        rng = rng or np.random.default_rng()
        # weights are registered in the Keras get_weights() order; the forward pass addresses them by layer name
        where = {}
        for name, kind, cin, cout, k in resnet50_layers():
            where[name] = len(self._entries)
            if kind == "conv":
                self.add_weight(name + "/kernel", he_normal(rng, (k, k, cin, cout)))
                self.add_weight(name + "/bias", np.zeros(cout, np.float32))
            else:
                # a residual branch's last BN starts small so 16 stacked blocks keep O(1) activations
                self.add_weight(name + "/gamma", np.full(cout, 0.25 if name.endswith("_3_bn") else 1.0, np.float32))
                self.add_weight(name + "/beta", np.zeros(cout, np.float32))
                self.add_weight(name + "/moving_mean", np.zeros(cout, np.float32), trainable=False)
                self.add_weight(name + "/moving_variance", np.ones(cout, np.float32), trainable=False)
        # execution order: (kernel index, bn index, spec) per conv+BN pair -- shortcut (0), then 1, 2, 3 of every block
        self._convs = []

        def conv_bn(prefix, spec):
            self._convs.append((where[prefix + "_conv"], where[prefix + "_bn"], spec))

        conv_bn("conv1", C7)
        for si, (filters, blocks, stride1) in enumerate(RESNET50_STACKS):
            for bi in range(blocks):
                n = "conv%d_block%d" % (si + 2, bi + 1)
                s = stride1 if bi == 0 else 1
                if bi == 0:
                    conv_bn(n + "_0", C1[s])                     # 0_conv shortcut
                conv_bn(n + "_1", C1[s])                         # 1_conv (stride on the first 1x1)
                conv_bn(n + "_2", C3)                            # 2_conv
                conv_bn(n + "_3", C1[1])                         # 3_conv
        self.resnet_feature_dim = 2048
        self.add_weight("rotation_regressor/kernel", glorot_uniform(rng, (2048, 3)))
        self.add_weight("rotation_regressor/bias", np.zeros(3, np.float32))
        self.add_weight("feature_to_latent_mlp/kernel", glorot_uniform(rng, (2048, latent_dim)))
        self.add_weight("feature_to_latent_mlp/bias", np.zeros(latent_dim, np.float32))
        self.finalize()
        mult = np.pi * np.array([rotation_ranges[0][1], rotation_ranges[1][1], rotation_ranges[2][1]]) / 180.0
        self.rotation_range_multiplier = torch.tensor(mult, dtype=torch.float32, device=self.device)

    def _bn_coefficients(self, cat=False):
        """Per-channel affine of every conv+BN pair, computed for all 53 pairs at once (3 concatenations and
        5 elementwise launches on ~53k channels instead of ~5 launches per pair):
        bn(conv + b) = a*conv + shift,  a = gamma*rsqrt(var+eps),  shift = beta + a*(b - mean).
        The (C,) coefficient algebra is host-side plumbing and carries the gradients of gamma, beta and b."""
        ws = self.weights
        ks = [k for k, _, _ in self._convs]
        bs = [b for _, b, _ in self._convs]
        sizes = [ws[k + 1].shape[0] for k in ks]
        bias = torch.cat([ws[k + 1] for k in ks])
        gamma = torch.cat([ws[b] for b in bs])
        beta = torch.cat([ws[b + 1] for b in bs])
        if getattr(self, "_stat_cache", None) is None:
            self._stat_cache = (torch.cat([ws[b + 2] for b in bs]), torch.cat([ws[b + 3] for b in bs]))
        mean, var = self._stat_cache
        a = gamma * torch.rsqrt(var + BN_EPS)
        shift = beta + a * (bias - mean)
        if cat:
            return a, torch.split(shift, sizes)
        return torch.split(a, sizes), torch.split(shift, sizes)

    def non_trainable_changed(self):
        """set_weights / copy_weights_from / a data-parallel broadcast rewrote the moving mean / variance in place: refill
        their concatenated copy IN PLACE -- captured step graphs read it at its address, so it is never dropped once made.
        (Not tied to mark_updated(): the optimizer calls that every step, and a copy re-made inside one captured step graph
        must not be shared with another.)"""
        cache = getattr(self, "_stat_cache", None)
        if cache is not None:
            bs = [b for _, b, _ in self._convs]
            with torch.no_grad():
                cache[0].copy_(torch.cat([self.weights[b + 2] for b in bs]))
                cache[1].copy_(torch.cat([self.weights[b + 3] for b in bs]))

    # ---- inference form (no tape): BatchNorm folded into the filters, bias / residual / ReLU in the convolutions' epilogues -----
    def _fold_table(self):
        """(int32 (53, 5) device table, packed size, [(packed offset, shape)]) for cn_scale_columns_segments: where each conv
        kernel lies in the weight arena, where its folded copy goes, its cout and where its BN coefficients start."""
        if getattr(self, "_fold_tab", None) is None:
            rows, views, dst, aoff = [], [], 0, 0
            base = self.arena.data_ptr()
            for kidx, _, _ in self._convs:
                w = self.weights[kidx]
                src = (w.data_ptr() - base) // 4
                assert 0 <= src and src % 4 == 0 and w.numel() % 4 == 0 and w.shape[-1] % 4 == 0 and w.is_contiguous()
                rows.append((src, dst, w.numel(), w.shape[-1], aoff))
                views.append((dst, tuple(w.shape)))
                dst += w.numel()
                aoff += w.shape[-1]
            self._fold_tab = (torch.tensor(rows, dtype=torch.int32, device=self.device), dst, views)
        return self._fold_tab

    def _folded_filters(self, a_cat):
        """Every conv kernel times its BatchNorm scale, in one launch, cached like any derived filter copy: per weight epoch of
        this network, per stream, and re-made INSIDE a HIP-graph capture (nn.WEIGHTS_EPOCH) so that a replay folds the weights
        of its own iteration."""
        from ..nn import WEIGHTS_EPOCH
        from .. import ops
        stream = torch.cuda.current_stream().cuda_stream
        key = (WEIGHTS_EPOCH[0], self.epoch)
        cache = self.__dict__.setdefault("_fold_cache", {})            # one entry per stream (two streams alternating must not re-fold)
        c = cache.get(stream)
        if c is None or c[0] != key:
            for k in [k for k, v in cache.items() if v[0] != key]:     # copies of older weights (other streams' included) are dead:
                del cache[k]                                           # the dict holds at most one packed copy per LIVE stream
            seg, total, views = self._fold_table()
            packed = ops.scale_columns_segments(self.arena, seg, a_cat, total)
            c = cache[stream] = (key, [packed[o:o + int(np.prod(shp))].view(shp) for o, shp in views], packed)
        if ops._keepalive is not None:                                 # on a cache hit too: a graph captured now reads `packed`
            ops._keepalive.append(c[2])
        return c[1]

    def _features_folded(self, img):
        """features() without a tape (the encoder of the discriminator-type steps, predict()): conv -> BN -> ReLU as ONE launch per
        layer -- bn(conv(x, w) + b) = conv(x, w * a) + shift with the coefficients of _bn_coefficients -- and the block's
        Add + ReLU in the epilogue of its last convolution (ops.conv_fwd_res)."""
        from .. import ops
        from ..ops import ACT_NONE, ACT_RELU
        a, shift = self._bn_coefficients(cat=True)
        wf = self._folded_filters(a)

        def conv(ci, x, res=None, relu=True):
            spec = self._convs[ci][2]
            g = spec.geom(tuple(x.shape), wf[ci].shape[-1])
            if res is None:
                return ops.conv_fwd(x, wf[ci], shift[ci], g, ACT_RELU if relu else ACT_NONE)
            return ops.conv_fwd_res(x, wf[ci], shift[ci], res, g, ACT_RELU)

        x = F.caffe_preprocess(img)
        x = conv(0, x)
        x = F.maxpool(x, 3, 2, 1)
        ci = 1
        for filters, blocks, stride1 in RESNET50_STACKS:
            for bi in range(blocks):
                if bi == 0:
                    sc = conv(ci, x, relu=False)
                    ci += 1
                else:
                    sc = x
                y = conv(ci, x)
                y = conv(ci + 1, y)
                x = conv(ci + 2, y, res=sc)
                ci += 3
        return F.global_avg_pool(x)

    def _conv_bn(self, ci, x, coef, res=None, relu=True):
        kidx, _, spec = self._convs[ci]
        z = F.conv(x, self.weights[kidx], None, spec)           # bias folded into the affine shift
        return F.channel_affine_act(z, coef[0][ci], coef[1][ci], res, relu)

    fold_inference = True      # (False: the taped form in every mode -- cross-check)

    def features(self, img):
        if self.fold_inference and not torch.is_grad_enabled() and img.dtype == torch.float32:
            from .. import ops
            if ops.ACT_DTYPE == torch.float32:
                return self._features_folded(img)
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
