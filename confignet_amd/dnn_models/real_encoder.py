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

    # ---- taped form on the folded filters (round 6) ---------------------------------------------------------------------------
    def _fold_table9(self):
        """(int32 (53, 9) device table, workgroups) for cn_bn_fold_bwd: the five columns of _fold_table, then the arena offsets of
        the layer's bias, gamma and beta, then the layer's first workgroup (one per 64 output channels)."""
        if getattr(self, "_fold_tab9", None) is None:
            seg = self._fold_table()[0].cpu().numpy()
            base = self.arena.data_ptr()
            rows, blk = [], 0
            for (kidx, bidx, _), r in zip(self._convs, seg):
                cout = int(r[3])
                assert cout % 64 == 0
                offs = [(self.weights[i].data_ptr() - base) // 4 for i in (kidx + 1, bidx, bidx + 1)]
                assert all(o >= 0 and o % 4 == 0 for o in offs)
                rows.append(list(map(int, r)) + offs + [blk])
                blk += cout // 64
            self._fold_tab9 = (torch.tensor(rows, dtype=torch.int32, device=self.device), blk)
        return self._fold_tab9

    def _fold_coefficients(self):
        """(a, shift, rs, bm) concatenated over the 53 conv + BN pairs, off the tape: a = gamma rs, rs = rsqrt(var + eps) (a
        constant: the moving variance is never updated), bm = bias - mean, shift = beta + a bm."""
        ws = self.weights
        ks = [k for k, _, _ in self._convs]
        bs = [b for _, b, _ in self._convs]
        with torch.no_grad():
            if getattr(self, "_stat_cache", None) is None:
                self._stat_cache = (torch.cat([ws[b + 2] for b in bs]), torch.cat([ws[b + 3] for b in bs]))
            mean, var = self._stat_cache
            rs = torch.rsqrt(var + BN_EPS)
            a = torch.cat([ws[b] for b in bs]) * rs
            bm = torch.cat([ws[k + 1] for k in ks]) - mean
            shift = torch.addcmul(torch.cat([ws[b + 1] for b in bs]), a, bm)
        return a, shift, rs, bm

    def _trunk_params(self):
        """The trainable tensors of the ResNet-50 trunk (everything but the two heads), in weight order."""
        return [w for w in self.weights[:-4] if w.requires_grad]

    folded_tape = True         # (False: the taped form as convolution + per-channel affine pass per layer -- cross-check)

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
        from .. import ops as _ops
        if self.folded_tape and torch.is_grad_enabled() and (_ops.ACT_DTYPE == torch.float32 or self.folded_tape == "always"):
            # (bf16 storage: the residual adds are passes of their own there -- no epilogue form -- and the folded filters need a second
            # derived copy; alternating pipelined runs: 752 images/s composite against 749 folded, fp32 417 against 420)
            params = self._trunk_params()
            if len(params) == sum(1 for i in self._trainable_idx if i < len(self.weights) - 4) or (not params and img.requires_grad):
                x = F.caffe_preprocess(img)
                return F.global_avg_pool(ResNetTrunkFn.apply(self, x, *params))
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


class ResNetTrunkFn(torch.autograd.Function):
    """The ResNet-50 trunk (conv1 .. conv5_block3_out) on the tape as ONE node, forward AND backward on the FOLDED filters
    (round 6).  BatchNormalization runs in inference mode (R9), so  relu(bn(conv(x, w) + b) [+ shortcut])  is
    relu(conv(x, w a) + shift [+ shortcut])  with a = gamma rsqrt(var + eps), shift = beta + a (b - mean): one launch per
    layer with bias / residual / ReLU in its epilogue -- the form the tape-free encoder has used since round 4 -- instead of a
    convolution plus a per-channel affine pass over its output.  Backward per layer: ReLU mask + shift sums in one pass
    (cn_act_bwd_bias), data gradient (a block's skip gradient added in the epilogue of its first convolution's data gradient,
    cn_conv_dgrad_w_res, instead of by autograd's add), filter gradient WRITTEN into a packed scratch; the gradients of
    kernel / bias / gamma / beta of all 53 pairs then come out of ONE launch at the join of the pass (cn_bn_fold_bwd):
    d gamma needs sum_k g'[k][c] w[k][c] over the small filter, not a reduction of g z over the activation tensor, so the
    pre-affine convolution output z is neither kept nor read.  Same function and gradients as the composite form
    (RealEncoder.folded_tape = False) up to summation order.  First-order only."""

    @staticmethod
    def forward(ctx, enc, x, *params):
        from .. import ops
        from ..ops import ACT_NONE, ACT_RELU
        a, shift, rs, bm = enc._fold_coefficients()
        wf = enc._folded_filters(a)
        sizes = [wf[ci].shape[-1] for ci in range(len(enc._convs))]
        offs = [0]
        for c in sizes:
            offs.append(offs[-1] + c)
        sh = [shift[offs[i]:offs[i + 1]] for i in range(len(sizes))]
        x = x.contiguous()

        def conv(ci, t, res=None, relu=True):
            g = enc._convs[ci][2].geom(tuple(t.shape), sizes[ci])
            if res is None:
                return ops.conv_fwd(t, wf[ci], sh[ci], g, ACT_RELU if relu else ACT_NONE)
            return ops.conv_fwd_res(t, wf[ci], sh[ci], res, g, ACT_RELU)

        saved = [x]
        y0 = conv(0, x)
        t = ops.maxpool_fwd(y0, 3, 2, 1)
        saved += [y0, t]
        ci = 1
        for filters, blocks, stride1 in RESNET50_STACKS:
            for bi in range(blocks):
                sc = t
                if bi == 0:
                    sc = conv(ci, t, relu=False)
                    ci += 1
                y1 = conv(ci, t)
                y2 = conv(ci + 1, y1)
                t = conv(ci + 2, y2, res=sc)
                saved += [y1, y2, t]
                ci += 3
        ctx.enc, ctx.coef, ctx.wf, ctx.offs = enc, (a, rs, bm), wf, offs
        ctx.save_for_backward(*saved)
        return t

    @staticmethod
    def backward(ctx, gy):
        from .. import ops
        from ..ops import ACT_RELU
        if torch.is_grad_enabled():
            raise RuntimeError("ResNetTrunkFn is first-order only")
        enc, wf, offs = ctx.enc, ctx.wf, ctx.offs
        a, rs, bm = ctx.coef
        saved = list(ctx.saved_tensors)
        seg5, total, views = enc._fold_table()
        seg9, blocks = enc._fold_table9()
        dev = gy.device
        params = enc._trunk_params()
        want_w = any(ctx.needs_input_grad[2:]) and not F._INPUT_GRADS_ONLY
        sunk = want_w and ops.sink_for(params[0]) is not None
        # (a pass whose sink does not hold this network: nothing of this node may be deferred to that sink's join)
        defer_prev, ops.DEFER_SLAB_SUMS = ops.DEFER_SLAB_SUMS, ops.DEFER_SLAB_SUMS and sunk
        gwf = gsh = None
        if want_w:
            gwf = torch.empty(total, device=dev, dtype=torch.float32)    # every layer WRITES its part (accumulate=False)
            gsh = ops.zero_pool_alloc((offs[-1],), dev)
            if gsh is None:
                gsh = torch.zeros(offs[-1], device=dev, dtype=torch.float32)

        def geom(ci, t):
            return enc._convs[ci][2].geom(tuple(t.shape), wf[ci].shape[-1])

        def wgrad(ci, t_in, gu):
            o, shp = views[ci]
            if want_w:
                ops.sink_conv_wgrad_to(t_in, gu, geom(ci, t_in), shp, gwf[o:o + int(np.prod(shp))].view(shp))

        def relu_bwd(ci, g, y, also=None):
            """g * (y > 0) and its channel sums into the shift gradient of layer ci (and of layer `also`: a block's shortcut
            convolution sees the same gradient as its last one)."""
            if not want_w:
                return ops.act_bwd(g, y, ACT_RELU)
            gu, part = ops.act_bwd_partials(g, y, ACT_RELU)
            ops.sum_rows_into(part, gsh[offs[ci]:offs[ci + 1]])
            if also is not None:
                ops.sum_rows_into(part, gsh[offs[also]:offs[also + 1]])
            return gu

        try:
            return ResNetTrunkFn._walk(ctx, gy, enc, wf, offs, saved, views, gwf, gsh, want_w, sunk, params, geom, wgrad, relu_bwd,
                                       seg9, blocks, a, rs, bm)
        finally:
            ops.DEFER_SLAB_SUMS = defer_prev

    @staticmethod
    def _walk(ctx, gy, enc, wf, offs, saved, views, gwf, gsh, want_w, sunk, params, geom, wgrad, relu_bwd, seg9, blocks, a, rs, bm):
        from .. import ops
        # walk the blocks backwards; `pos` indexes the saved activations
        g = gy.contiguous()
        layout = []                                  # (first conv index of the block, has conv shortcut) in forward order
        ci = 1
        for filters, blocks_, stride1 in RESNET50_STACKS:
            for bi in range(blocks_):
                layout.append((ci, bi == 0))
                ci += 4 if bi == 0 else 3
        pos = len(saved)
        for first, has_sc in reversed(layout):
            y1, y2, t_out = saved[pos - 3], saved[pos - 2], saved[pos - 1]
            pos -= 3
            t_in = saved[pos - 1]
            c1 = first + 1 if has_sc else first
            gu3 = relu_bwd(c1 + 2, g, t_out, also=first if has_sc else None)
            wgrad(c1 + 2, y2, gu3)
            gu2 = relu_bwd(c1 + 1, ops.conv_dgrad(gu3, wf[c1 + 2], geom(c1 + 2, y2)), y2)
            wgrad(c1 + 1, y1, gu2)
            gu1 = relu_bwd(c1, ops.conv_dgrad(gu2, wf[c1 + 1], geom(c1 + 1, y1)), y1)
            wgrad(c1, t_in, gu1)
            if has_sc:
                wgrad(first, t_in, gu3)
                skip = ops.conv_dgrad(gu3, wf[first], geom(first, t_in))
            else:
                skip = gu3
            g = ops.conv_dgrad_res(gu1, wf[c1], geom(c1, t_in), skip)
        x, y0, t = saved[0], saved[1], saved[2]
        g = ops.maxpool_bwd(y0, g, 3, 2, 1)
        gu0 = relu_bwd(0, g, y0)
        wgrad(0, x, gu0)
        gx = ops.conv_dgrad(gu0, wf[0], geom(0, x)) if ctx.needs_input_grad[1] else None
        if not want_w:
            return (None, gx) + (None,) * len(params)
        if sunk:
            gout = enc.grad_arena
            ops.sink_post(lambda: ops.bn_fold_bwd(seg9, blocks, gwf, gsh, enc.arena, a, rs, bm, gout), keep=(gwf, gsh, a, rs, bm))
            return (None, gx) + (None,) * len(params)
        gout = torch.zeros_like(enc.arena)
        ops.bn_fold_bwd(seg9, blocks, gwf, gsh, enc.arena, a, rs, bm, gout)
        base = enc.arena.data_ptr()
        grads = []
        for p_ in params:
            o = (p_.data_ptr() - base) // 4
            grads.append(gout[o:o + p_.numel()].view(p_.shape))
        return (None, gx) + tuple(grads)
