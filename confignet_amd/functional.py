"""Differentiable ops: torch.autograd.Function wrappers whose forward AND backward are calls
into libconfignet_hip.so.  torch.autograd is only the tape driver (the reference uses
tf.GradientTape the same way); no torch compute kernel runs on activation-sized tensors.

Double backward (the R1 penalty, losses.py:75-82, differentiates an input-gradient) is obtained
by writing every backward in terms of other Functions of this module: the sets
{conv, conv_dgrad, conv_wgrad}, {matmul}, {nc_sumdot, nc_lin} and {lrelu, lrelu_mask_mul} are
each closed under differentiation.
"""
import contextlib

import torch
from torch.autograd import Function

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, ConvSpec  # noqa: F401

_INPUT_GRADS_ONLY = False
FUSED_TAIL_STATS = True    # DiscrBlock tail: the two statistics passes as one (cn_nc_reduce4)
FUSED_R1_TAIL = True     # tangent tail backward (DualTailBatchedFn): cn_nc_reduce_hxt + cn_dual_tail_gx_tx (two passes over h instead of five launches)
BN_BWD_FUSED = True      # conv -> BN(inference) -> ReLU backward as one pass (cn_bn_act_bwd)


@contextlib.contextmanager
def input_grads_only():
    """Inside this context backward passes skip parameter gradients (used while taking
    d out / d input for the R1 penalty, where tf's tape.gradient(out, real_imgs) computes
    nothing else either)."""
    global _INPUT_GRADS_ONLY
    prev, _INPUT_GRADS_ONLY = _INPUT_GRADS_ONLY, True
    try:
        yield
    finally:
        _INPUT_GRADS_ONLY = prev


def _cg(t):
    return t if t.is_contiguous() else t.contiguous()


# =============================================================================================
# convolution family
# =============================================================================================
class ConvFn(Function):
    """y = act(conv(x [upsampled x2], w) + bias)."""

    @staticmethod
    def forward(ctx, x, w, bias, spec, act, slope):
        x, w = _cg(x), _cg(w)
        g = spec.geom(tuple(x.shape), w.shape[-1])
        if ops.upfold_ok(g):
            # UpSampling + Conv collapsed per output-parity class: 8/27 (3-D k3) or 6.25/16 (2-D k4) of the multiply-adds
            wf, _, gd, _ = ops.upfold_prepare(w, g)
            y = ops.conv_fwd(x, wf, bias, gd, act, slope)
            ops.prof_note_saved(ops.upfold_saved_flops(g))
        else:
            y = ops.conv_fwd(x, w, bias, g, act, slope)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.g, ctx.act, ctx.slope, ctx.has_bias = g, act, slope, bias is not None
        ctx.bias_ref = bias if (bias is not None and bias.requires_grad) else None     # (only its identity: gradient-sink lookup)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = _cg(gy)
        gb_fused = None
        first_order = not torch.is_grad_enabled()
        want_w = ctx.needs_input_grad[1] and not _INPUT_GRADS_ONLY
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and not _INPUT_GRADS_ONLY
        # inside nn.backward_into_arenas: filter / bias gradients are added straight into their gradient-arena slots by the
        # kernels, on the sink's side stream (ops.grad_sink); autograd then sees None for them
        w_slot = ops.sink_for(w) if (first_order and want_w) else None
        b_slot = ops.sink_for(ctx.bias_ref) if (first_order and want_b and ctx.bias_ref is not None) else None
        if ctx.act != ACT_NONE:
            if not first_order:
                raise RuntimeError("double backward through a fused-activation conv is not supported; "
                                   "use conv(..., act=ACT_NONE) + lrelu()")
            if want_b:
                gy, gb_fused = ops.act_bwd_bias(gy, y, ctx.act, ctx.slope, sink=b_slot)   # one pass: activation gradient + its channel sums
            else:
                gy = ops.act_bwd(gy, y, ctx.act, ctx.slope)
        elif want_b and b_slot is not None:
            ops.bias_grad(gy.detach(), sink=b_slot)
        gx = gw = gb = None
        if want_b and b_slot is None:
            gb = gb_fused if gb_fused is not None else ops.bias_grad(gy.detach())
        if ops.upfold_ok(ctx.g) and first_order:
            # first-order backward of the collapsed form: data gradient straight at the stored extent (no upsampled gradient,
            # no sum-pool pass), filter gradient of the class filters scattered back to the k taps
            _, wd, _, g2 = ops.upfold_prepare(w, ctx.g)
            if ctx.needs_input_grad[0]:
                gx = ops.conv_fwd(gy, wd, None, g2)
                ops.prof_note_saved(ops.upfold_saved_flops(ctx.g))
            if want_w:
                if w_slot is not None:
                    ops.sink_upfold_wgrad(gy, x, g2, tuple(wd.shape), ctx.g, tuple(w.shape), w_slot)
                else:
                    gw = ops.upfold_wgrad(ops.conv_wgrad(gy, x, g2, tuple(wd.shape)), ctx.g, tuple(w.shape))
                ops.prof_note_saved(ops.upfold_saved_flops(ctx.g))
            return gx, gw, gb, None, None, None
        if ctx.needs_input_grad[0]:
            gx = ConvDgradFn.apply(gy, w, ctx.g)
        if want_w:
            if w_slot is not None:
                ops.sink_conv_wgrad(x, gy, ctx.g, tuple(w.shape), w_slot)
            else:
                gw = ConvWgradFn.apply(x, gy, ctx.g, tuple(w.shape))
        return gx, gw, gb, None, None, None


class ConvDgradFn(Function):
    """gx = d conv / d x applied to gy (includes the 2^nd-child sum of a folded upsample)."""

    @staticmethod
    def forward(ctx, gy, w, g):
        gy = _cg(gy)
        gu = ops.conv_dgrad(gy, w, g)
        ctx.save_for_backward(gy, w)
        ctx.g = g
        return ops.sumpool2(gu) if g.up else gu

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        ggx = _cg(ggx)
        g_gy = g_w = None
        if ctx.needs_input_grad[0]:
            g_gy = ConvNoBiasFromGeomFn.apply(ggx, w, ctx.g)
        if ctx.needs_input_grad[1]:
            g_w = ConvWgradFn.apply(ggx, gy, ctx.g, tuple(w.shape))
        return g_gy, g_w, None


class ConvNoBiasFromGeomFn(Function):
    """conv(x, w) for an already-built geometry (the adjoint of ConvDgradFn)."""

    @staticmethod
    def forward(ctx, x, w, g):
        x = _cg(x)
        ctx.save_for_backward(x, w)
        ctx.g = g
        return ops.conv_fwd(x, _cg(w), None, g, ACT_NONE, 0.0)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _cg(gy)
        gx = ConvDgradFn.apply(gy, w, ctx.g) if ctx.needs_input_grad[0] else None
        gw = ConvWgradFn.apply(x, gy, ctx.g, tuple(w.shape)) if ctx.needs_input_grad[1] else None
        return gx, gw, None


class ConvWgradFn(Function):
    @staticmethod
    def forward(ctx, x, gy, g, w_shape):
        x, gy = _cg(x), _cg(gy)
        ctx.save_for_backward(x, gy)
        ctx.g, ctx.w_shape = g, w_shape
        return ops.conv_wgrad(x, gy, g, w_shape)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        ggw = _cg(ggw)
        gx = ConvDgradFn.apply(gy, ggw, ctx.g) if ctx.needs_input_grad[0] else None
        ggy = ConvNoBiasFromGeomFn.apply(x, ggw, ctx.g) if ctx.needs_input_grad[1] else None
        return gx, ggy, None, None


def conv(x, w, bias, spec, act=ACT_NONE, slope=0.0):
    return ConvFn.apply(x, w, bias, spec, act, slope)


# =============================================================================================
# dense
# =============================================================================================
class MatmulFn(Function):
    """op(a) @ op(b); closed under differentiation."""

    @staticmethod
    def forward(ctx, a, b, ta, tb):
        a, b = _cg(a), _cg(b)
        ctx.save_for_backward(a, b)
        ctx.ta, ctx.tb = ta, tb
        return ops.gemm(a, b, ta, tb)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        ta, tb = ctx.ta, ctx.tb
        ga = gb = None
        if ctx.needs_input_grad[0]:
            # C = op(a) op(b): d/d op(a) = gc op(b)^T ; transpose back if ta
            ga = MatmulFn.apply(b, gc, tb, True) if ta else MatmulFn.apply(gc, b, False, not tb)
        if ctx.needs_input_grad[1] and not (_INPUT_GRADS_ONLY and b.is_leaf):
            gb = MatmulFn.apply(gc, a, True, ta) if tb else MatmulFn.apply(a, gc, not ta, False)
        return ga, gb, None, None


class LinearFn(Function):
    """Keras Dense without activation: y = x @ W + b (bias fused into the GEMM epilogue)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _cg(x), _cg(w)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.bias_ref = b if (b is not None and b.requires_grad) else None
        return ops.gemm(x, w, False, False, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _cg(gy)
        gx = gw = gb = None
        first_order = not torch.is_grad_enabled()
        if ctx.needs_input_grad[0]:
            gx = MatmulFn.apply(gy, w, False, True)
        if not _INPUT_GRADS_ONLY:
            if ctx.needs_input_grad[1]:
                slot = ops.sink_for(w) if first_order else None
                if slot is not None:
                    ops.sink_gemm(x, gy, slot, True, False)
                else:
                    gw = MatmulFn.apply(x, gy, True, False)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                slot = ops.sink_for(ctx.bias_ref) if (first_order and ctx.bias_ref is not None) else None
                gb = ops.bias_grad(gy.detach(), sink=slot)
        return gx, gw, gb


class LinearActFn(Function):
    """Dense + fused activation; first-order only (generator / encoder MLPs)."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope):
        x, w = _cg(x), _cg(w)
        y = ops.gemm(x, w, False, False, b, act, slope)
        ctx.save_for_backward(x, w, y)
        ctx.act, ctx.slope, ctx.has_bias = act, slope, b is not None
        ctx.bias_ref = b if (b is not None and b.requires_grad) else None
        return y

    @staticmethod
    def backward(ctx, gy):
        if torch.is_grad_enabled():
            raise RuntimeError("LinearActFn is first-order only; use linear() + lrelu()")
        x, w, y = ctx.saved_tensors
        gb = gw = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g, gb = ops.act_bwd_bias(_cg(gy), y, ctx.act, ctx.slope, sink=ops.sink_for(ctx.bias_ref))
        else:
            g = ops.act_bwd(_cg(gy), y, ctx.act, ctx.slope)
        gx = ops.gemm(g, w, False, True) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[1]:
            slot = ops.sink_for(w)
            if slot is not None:
                ops.sink_gemm(x, g, slot, True, False)
            else:
                gw = ops.gemm(x, g, True, False)
        return gx, gw, gb, None, None


def linear(x, w, b=None, act=ACT_NONE, slope=0.0):
    if act == ACT_NONE:
        return LinearFn.apply(x, w, b)
    return LinearActFn.apply(x, w, b, act, slope)


class MlpBankFn(Function):
    """Several two-layer MLPs (Dense -> LeakyReLU(slope) -> Dense, building_blocks.py:152-173) as ONE tape node with one grouped
    launch per layer (cn_gemm_rows_grouped): the AdaIN MLPs of a generator pass (hologan_generator.py:119-124) -- 12 launches
    forward and ~24 backward become 2 + 2; the gradients of inputs that are the SAME tensor are added by the launch (atomics,
    <= 6 terms per element; separate outputs + autograd's add in deterministic mode).  The node runs backward once the
    cotangents of all its outputs have arrived, i.e. at the end of the generator's backward pass -- which is where the latent's
    gradient is first needed.  Inputs: slope, n, then z_0 .. z_{n-1}, then (w0, b0, w1, b1) per MLP.  First-order only."""

    @staticmethod
    def forward(ctx, slope, n, *ts):
        ctx.set_materialize_grads(False)
        zs = [_cg(z) for z in ts[:n]]
        ws = ts[n:]
        hs, outs, j1, j2 = [], [], [], []
        for i, z in enumerate(zs):
            w0, b0, w1, b1 = ws[4 * i:4 * i + 4]
            h = torch.empty((z.shape[0], w0.shape[1]), device=z.device, dtype=torch.float32)
            o = torch.empty((z.shape[0], w1.shape[1]), device=z.device, dtype=torch.float32)
            j1.append((z, w0, h, b0, None, False, ACT_LRELU, slope, False))
            j2.append((h, w1, o, b1, None, False, ACT_NONE, 0.0, False))
            hs.append(h)
            outs.append(o)
        ops.gemm_rows_grouped(j1)
        ops.gemm_rows_grouped(j2)
        ctx.save_for_backward(*zs, *hs)
        ctx.ws, ctx.n, ctx.slope = ws, n, slope
        # the first input that is this very tensor (same storage, shape and strides: views of one tensor feed one gradient)
        ctx.same = [next(k for k in range(i + 1) if ts[k].data_ptr() == ts[i].data_ptr() and ts[k].shape == ts[i].shape
                         and ts[k].stride() == ts[i].stride() and ts[k].dtype == ts[i].dtype) for i in range(n)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gos):
        if torch.is_grad_enabled():
            raise RuntimeError("MlpBankFn is first-order only")
        n, ws, slope = ctx.n, ctx.ws, ctx.slope
        saved = ctx.saved_tensors
        zs, hs = saved[:n], saved[n:]
        live = [i for i in range(n) if gos[i] is not None]
        gos = [None if g is None else _cg(g).float() for g in gos]
        need_z = [ctx.needs_input_grad[2 + i] for i in range(n)]
        ghs, j1, j2 = {}, [], []
        for i in live:
            gh = torch.empty_like(hs[i])
            ghs[i] = gh
            ops._log_mask(hs[i], ACT_LRELU)                                                           # (test instrument: ops.branch_log)
            j1.append((gos[i], ws[4 * i + 2], gh, None, hs[i], True, ACT_LRELU, slope, False))       # (go w1^T) * lrelu'(h)
        if j1:
            ops.gemm_rows_grouped(j1)
        gzs, shared = [None] * n, {}
        merged = not ops.DETERMINISTIC
        for i in live:
            if not need_z[i]:
                continue
            k = ctx.same[i] if merged else i
            if gzs[k] is None:
                shared[k] = merged and sum(1 for i2 in live if ctx.same[i2] == k and need_z[i2]) > 1
                if shared[k]:
                    gzs[k] = ops.zero_pool_alloc(tuple(zs[i].shape), zs[i].device)
                    if gzs[k] is None:
                        gzs[k] = torch.zeros_like(zs[i])
                else:
                    gzs[k] = torch.empty_like(zs[i])
            j2.append((ghs[i], ws[4 * i], gzs[k], None, None, True, ACT_NONE, 0.0, shared[k]))
        if j2:
            ops.gemm_rows_grouped(j2)
        grads = [None, None] + gzs
        for i in range(n):
            w0, b0, w1, b1 = ws[4 * i:4 * i + 4]
            gw = [None] * 4
            if i in ghs and not _INPUT_GRADS_ONLY:
                for (w, x_, g_, slot_i) in ((w0, zs[i], ghs[i], 0), (w1, hs[i], gos[i], 2)):
                    if ctx.needs_input_grad[2 + n + 4 * i + slot_i]:
                        slot = ops.sink_for(w)
                        if slot is not None:
                            ops.sink_gemm(x_, g_, slot, True, False)
                        else:
                            gw[slot_i] = ops.gemm(x_, g_, True, False)
                for (b, g_, slot_i) in ((b0, ghs[i], 1), (b1, gos[i], 3)):
                    if b is not None and ctx.needs_input_grad[2 + n + 4 * i + slot_i]:
                        gw[slot_i] = ops.bias_grad(g_, sink=ops.sink_for(b))
            grads += gw
        return tuple(grads)


def mlp_bank(zs, weight_sets, slope):
    """[MLP_i(z_i)] for two-layer MLPs with weights (w0, b0, w1, b1)_i: one grouped launch per layer (MlpBankFn)."""
    flat = [t for ws in weight_sets for t in ws]
    return list(MlpBankFn.apply(slope, len(zs), *zs, *flat))


# =============================================================================================
# activations
# =============================================================================================
class LreluFn(Function):
    @staticmethod
    def forward(ctx, x, slope):
        y = ops.act_fwd(_cg(x), ACT_LRELU, slope)
        ctx.save_for_backward(y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return LreluMaskMulFn.apply(gy, y, ctx.slope), None


class LreluMaskMulFn(Function):
    """g * lrelu'(.) with the mask taken from the sign of y (sign(y) == sign(x), slope > 0)."""

    @staticmethod
    def forward(ctx, g, y, slope):
        ctx.save_for_backward(y)
        ctx.slope = slope
        return ops.act_bwd(_cg(g), y, ACT_LRELU, slope)

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return LreluMaskMulFn.apply(gg, y, ctx.slope), None, None


def lrelu(x, slope):
    return LreluFn.apply(x, slope)


# =============================================================================================
# per-(sample, channel) statistics / affine maps on channels-last tensors
# =============================================================================================
class NcSumDotFn(Function):
    """(sum_s a, sum_s a*b) per (n, c) [per c when per_channel]; b=None means b = a."""

    @staticmethod
    def forward(ctx, a, b, per_channel):
        a = _cg(a)
        b = _cg(b) if b is not None else None
        ctx.save_for_backward(a, b)
        ctx.per_channel = per_channel
        s1, s2 = ops.nc_reduce(a, b, per_channel=per_channel)
        return s1, s2

    @staticmethod
    def backward(ctx, g1, g2):
        a, b = ctx.saved_tensors
        pc = ctx.per_channel
        ga = gb = None
        g1 = _cg(g1) if g1 is not None else None
        g2 = _cg(g2) if g2 is not None else None
        if b is None:
            if ctx.needs_input_grad[0]:
                ga = NcLinFn.apply(a, None if g2 is None else 2.0 * g2, None, None, g1, pc) if g2 is not None else \
                    NcLinFn.apply(None, None, None, None, g1, pc, tuple(a.shape))
        else:
            if ctx.needs_input_grad[0]:
                ga = NcLinFn.apply(b, g2, None, None, g1, pc) if g2 is not None else \
                    NcLinFn.apply(None, None, None, None, g1, pc, tuple(a.shape))
            if ctx.needs_input_grad[1] and g2 is not None:
                gb = NcLinFn.apply(a, g2, None, None, None, pc)
        return ga, gb, None


class NcLinFn(Function):
    """y = a1*x1 + a2*x2 + b, coefficients (n, c) [or (1, c) when per_channel] broadcast over space.
    Any of (x1,a1), (x2,a2), b may be None; a missing coefficient of a present x means 1."""

    @staticmethod
    def forward(ctx, x1, a1, x2, a2, b, per_channel, shape=None):
        x1 = _cg(x1) if x1 is not None else None
        x2 = _cg(x2) if x2 is not None else None
        a1 = _cg(a1) if a1 is not None else None
        a2 = _cg(a2) if a2 is not None else None
        b = _cg(b) if b is not None else None
        if shape is None:
            shape = tuple((x1 if x1 is not None else x2).shape)
        ctx.save_for_backward(x1, a1, x2, a2)
        ctx.per_channel, ctx.has_b = per_channel, b is not None
        return ops.nc_lin2(shape, x1, a1, x2, a2, b, per_channel=per_channel)

    @staticmethod
    def backward(ctx, gy):
        x1, a1, x2, a2 = ctx.saved_tensors
        pc = ctx.per_channel
        gy = _cg(gy)
        gx1 = ga1 = gx2 = ga2 = gb = None
        need_b = ctx.has_b and ctx.needs_input_grad[4]
        sum_gy = None
        if x1 is not None and ctx.needs_input_grad[0]:
            gx1 = NcLinFn.apply(gy, a1, None, None, None, pc)
        if a1 is not None and ctx.needs_input_grad[1]:
            sum_gy, ga1 = NcSumDotFn.apply(gy, x1, pc)
        if x2 is not None and ctx.needs_input_grad[2]:
            gx2 = NcLinFn.apply(gy, a2, None, None, None, pc)
        if a2 is not None and ctx.needs_input_grad[3]:
            sum_gy2, ga2 = NcSumDotFn.apply(gy, x2, pc)
            sum_gy = sum_gy if sum_gy is not None else sum_gy2
        if need_b:
            gb = sum_gy if sum_gy is not None else NcSumDotFn.apply(gy, None, pc)[0]
        return gx1, ga1, gx2, ga2, gb, None, None


def nc_stats(x, per_channel=False):
    """(sum x, sum x^2) in one pass."""
    return NcSumDotFn.apply(x, None, per_channel)


def nc_lin(x1, a1=None, x2=None, a2=None, b=None, per_channel=False):
    return NcLinFn.apply(x1, a1, x2, a2, b, per_channel)


def _spatial(x):
    return x.numel() // (x.shape[0] * x.shape[-1])


class AdaInFn(Function):
    """AdaIn.call (building_blocks.py:135-149) fused: statistics pass, one coefficient kernel, one
    normalise+modulate pass; backward = one reduction pass + one coefficient kernel + one pass.
    First-order only (the generator is never differentiated twice)."""

    @staticmethod
    def forward(ctx, x, scale_bias):
        x, sb = _cg(x), _cg(scale_bias)
        sp = _spatial(x)
        st = ops.take_stats(x, "act")                # left by the producing convolution's epilogue (ops.request_stats), else one pass
        s1, s2 = st if st is not None else ops.nc_reduce(x)
        fused = ops.norm_apply_fwd(ops.NORM_ADAIN, x, s1, s2, sb, None, 1e-3)        # coefficients inline in the apply pass
        if fused is not None:
            y, mean, r = fused
            ctx.save_for_backward(x, sb, mean, r)
            return y
        a, b, mean, r = ops.norm_coef_fwd(ops.NORM_ADAIN, s1, s2, sb, None, sp, 1e-3)
        ctx.save_for_backward(x, sb, mean, r)
        return ops.nc_lin2(tuple(x.shape), x, a, b=b)

    @staticmethod
    def backward(ctx, gy):
        if torch.is_grad_enabled():
            raise RuntimeError("AdaInFn is first-order only; use adain_composite()")
        x, sb, mean, r = ctx.saved_tensors
        gy = _cg(gy)
        t1, t2 = ops.nc_reduce(gy, x)
        fused = ops.norm_apply_bwd(ops.NORM_ADAIN, gy, x, t1, t2, mean, r, sb, 1e-3)
        if fused is not None:
            return fused[0], fused[1]
        c1, c2, c0, gsb, _ = ops.norm_coef_bwd(ops.NORM_ADAIN, t1, t2, mean, r, sb, _spatial(x), 1e-3)
        return ops.nc_lin2(tuple(x.shape), gy, c1, x, c2, c0), gsb


def adain(x, scale_bias):
    """AdaIn.call (building_blocks.py:135-149): LayerNormalization over the spatial axes (eps 1e-3,
    no affine) then x*(s+1)+b; scale_bias (N, 2C) = [s | b] from the MLP of z."""
    return AdaInFn.apply(x, scale_bias)


def adain_composite(x, scale_bias):
    """Same maths from twice-differentiable primitives (kept as a cross-check of AdaInFn)."""
    c = x.shape[-1]
    s_, b_ = scale_bias[:, :c], scale_bias[:, c:]
    inv = 1.0 / _spatial(x)
    s1, s2 = nc_stats(x)
    mu = s1 * inv
    var = s2 * inv - mu * mu
    a = torch.rsqrt(var + 1e-3) * (s_ + 1.0)
    return nc_lin(x, a, None, None, b_ - mu * a)


class DiscrTailFn(Function):
    """Tail of DiscrBlock.call (building_blocks.py:100-106) fused, first-order only: style statistics of the
    pre-activation tensor, LeakyReLU, instance normalisation.  x is read by two reduction passes and one
    normalise pass; the backward pass is one reduction pass + one pass that also adds the style gradient.
    Returns (y, style|None, mean, q, style_mean|None, style_std|None); the statistics are non-differentiable
    side outputs reused by the tangent pass of the R1 penalty (DualTailFn)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, want_style, slope):
        ctx.set_materialize_grads(False)
        x, gamma, beta = _cg(x), _cg(gamma), _cg(beta)
        sp = _spatial(x)
        style = smean = ssd = None
        st = ops.take_stats(x, "pre4")               # (sum x, sum x^2, sum l, sum l^2) from the producing convolution's epilogue
        if st is not None:
            s1, s2, a1, a2 = st
            if want_style:
                style, _, smean, ssd = ops.norm_coef_fwd(ops.NORM_STYLE, s1, s2, None, None, sp, 1e-6)
        elif want_style and FUSED_TAIL_STATS and x.shape[-1] % 4 == 0:
            s1, s2, a1, a2 = ops.nc_reduce4(x, slope)          # style + instance-norm statistics: one pass over x instead of two
            style, _, smean, ssd = ops.norm_coef_fwd(ops.NORM_STYLE, s1, s2, None, None, sp, 1e-6)
        else:
            if want_style:
                s1, s2 = ops.nc_reduce(x)
                style, _, smean, ssd = ops.norm_coef_fwd(ops.NORM_STYLE, s1, s2, None, None, sp, 1e-6)
            a1, a2 = ops.nc_reduce(x, flags=1, slope=slope)
        fused = ops.norm_apply_fwd(ops.NORM_INSTANCE, x, a1, a2, gamma, beta, 1e-3, flags=1, slope=slope)
        if fused is not None:
            y, mean, q = fused
        else:
            a, b, mean, q = ops.norm_coef_fwd(ops.NORM_INSTANCE, a1, a2, gamma, beta, sp, 1e-3)
            y = ops.nc_lin2(tuple(x.shape), x, a, b=b, flags=1, slope=slope)
        ctx.save_for_backward(x, gamma, mean, q, smean, ssd)
        ctx.slope, ctx.want_style = slope, want_style
        nd = [t for t in (mean, q, smean, ssd) if t is not None]
        ctx.mark_non_differentiable(*nd)
        return y, style, mean, q, smean, ssd

    @staticmethod
    def backward(ctx, gy, gstyle, *_):
        if torch.is_grad_enabled():
            raise RuntimeError("DiscrTailFn is first-order only; use the composite discr_block path")
        x, gamma, mean, q, smean, ssd = ctx.saved_tensors
        sp = _spatial(x)
        have_y = gy is not None
        d2 = d0 = None
        if ctx.want_style and gstyle is not None:
            _, d2, d0, _, _ = ops.norm_coef_bwd(ops.NORM_STYLE, _cg(gstyle), None, smean, ssd, None, sp, 1e-6)
        if have_y:
            gy = _cg(gy)
            t1, t2 = ops.nc_reduce(gy, x, flags=2, slope=ctx.slope)
            fused = ops.norm_apply_bwd(ops.NORM_INSTANCE, gy, x, t1, t2, mean, q, gamma, 1e-3, flags=2 | 4, slope=ctx.slope, a3=d2, b3=d0)
            if fused is not None:
                return fused[0], fused[1], fused[2], None, None
            c1, c2, c0, ggamma, gbeta = ops.norm_coef_bwd(ops.NORM_INSTANCE, t1, t2, mean, q, gamma, sp, 1e-3)
            gx = ops.nc_lin2(tuple(x.shape), gy, c1, x, c2, c0, flags=2 | 4, slope=ctx.slope, a3=d2, b3=d0)
            return gx, ggamma, gbeta, None, None
        # only the style statistics carry gradient (R1 input-gradient of a style head)
        gx = ops.nc_lin2(tuple(x.shape), x, d2, b=d0) if d2 is not None else torch.zeros_like(x)
        return gx, torch.zeros_like(gamma), torch.zeros_like(gamma), None, None


class DualTailFn(Function):
    """Tangent of the DiscrBlock tail in direction tx (include/confignet_hip.h: cn_dual_tail_*): returns
    (ty | None, tstyle | None).  Its backward supplies the gradient w.r.t. the tangent input AND the
    second-order terms w.r.t. the primal pre-activation x -- this is what makes the R1 penalty trainable with
    ONE first-order backward pass."""

    @staticmethod
    def forward(ctx, tx, x, gamma, mean, q, smean, ssd, want_ty, want_style, slope):
        ctx.set_materialize_grads(False)
        tx, x = _cg(tx), _cg(x)
        sp = _spatial(x)
        shape = tuple(x.shape)
        ta = T = U = None
        if want_ty:
            ta = ops.act_bwd(tx, x, ACT_LRELU, slope)               # lrelu'(x) * tx
            T = ops.nc_reduce(ta, x, flags=2, slope=slope)          # sum ta, sum ta*lrelu(x)
        if want_style:
            U = ops.nc_reduce(tx, x)                                # sum tx, sum tx*x
        C1, C2, C0, tstyle = ops.dual_tail_coef_fwd(T, U, mean if want_ty else None, q if want_ty else None,
                                                    smean if want_style else None, ssd if want_style else None,
                                                    gamma, sp)
        ty = ops.nc_lin2(shape, ta, C1, x, C2, C0, flags=2, slope=slope) if want_ty else None
        ctx.save_for_backward(tx, x, gamma, mean, q, smean, ssd, ta,
                              T[0] if T else None, T[1] if T else None, U[0] if U else None, U[1] if U else None)
        ctx.cfg = (want_ty, want_style, slope)
        return ty, tstyle

    @staticmethod
    def backward(ctx, h, u):
        if torch.is_grad_enabled():
            raise RuntimeError("DualTailFn is first-order only")
        tx, x, gamma, mean, q, smean, ssd, ta, T1, T2, U1, U2 = ctx.saved_tensors
        want_ty, want_style, slope = ctx.cfg
        sp = _spatial(x)
        shape = tuple(x.shape)
        use_h = want_ty and h is not None
        use_u = want_style and u is not None
        H = E = None
        if use_h:
            h = _cg(h)
            H = ops.nc_reduce(h, x, flags=2, slope=slope)           # sum h, sum h*lrelu(x)
            E = ops.nc_reduce(h, ta, want_sum=False)[1]             # sum h*ta
        if not use_h and not use_u:
            return torch.zeros_like(tx), torch.zeros_like(x), torch.zeros_like(gamma), None, None, None, None, None, None, None
        co = ops.dual_tail_coef_bwd(H, E, _cg(u) if use_u else None, (T1, T2) if use_h else None,
                                    (U1, U2) if use_u else None, mean if use_h else None, q if use_h else None,
                                    smean if use_u else None, ssd if use_u else None, gamma, sp)
        if use_h:
            g_tx = ops.nc_lin2(shape, h, co["K1"], x, co["K2"], co["K0"], flags=2 | 4, slope=slope,
                               a3=co["D2"], b3=co["D0"])
        else:
            g_tx = ops.nc_lin2(shape, x, co["D2"], b=co["D0"])
        g_x = ops.dual_tail_gx(h if use_h else None, ta if use_h else None, tx, x, co, slope)
        g_gamma = co["ggamma"] if use_h else torch.zeros_like(gamma)
        return g_tx, g_x, g_gamma, None, None, None, None, None, None, None


class DualTailBatchedFn(Function):
    """DualTailFn for the tangents of SEVERAL heads at once (round 3): tx holds h*N samples, head-major -- the first N are the
    head that leaves the trunk here through this block's style statistics, the other (h-1)*N go on through LeakyReLU + instance
    norm -- against ONE copy of the primal pre-activation x (N samples, read through a sample period).  Returns
    (ty ((h-1)*N samples), tstyle (N, 2C)).  Same arithmetic as h calls of DualTailFn; the heads' second-order terms w.r.t. x
    are summed inside cn_dual_tail_gx.  With it the tangent pass of the R1 penalty (losses.py:75-82) is ONE convolution per
    block on the stacked tangents (and one data / filter gradient per block in its backward) instead of one per head."""

    @staticmethod
    def forward(ctx, tx, x, gamma, mean, q, smean, ssd, slope):
        ctx.set_materialize_grads(False)
        tx, x = _cg(tx), _cg(x)
        n = x.shape[0]
        assert tx.shape[0] % n == 0 and tx.shape[0] >= 2 * n and tx.shape[1:] == x.shape[1:]
        sp = _spatial(x)
        tx_s, tx_r = tx[:n], tx[n:]
        lazy = FUSED_R1_TAIL and x.shape[-1] % 4 == 0 and tx.dtype == x.dtype
        # ta = lrelu'(x) tx; sum ta, sum ta*lrelu(x).  lazy: ta is NOT stored -- every later reader forms it from tx and x (one write and
        # 5 N samples of activation memory per block less)
        ta, T1, T2 = ops.nc_reduce_dact(tx_r, x, ACT_LRELU, slope, x2_period=n, flags=2, want_a=not lazy)
        U = ops.nc_reduce(tx_s, x)                                                            # sum tx, sum tx*x
        C1, C2, C0, tstyle = ops.dual_tail_coef_fwd((T1, T2), U, mean, q, smean, ssd, gamma, sp)
        if lazy:
            ty = ops.nc_lin2(tuple(tx_r.shape), tx_r, C1, x, C2, C0, flags=2 | 16, slope=slope, x2_period=n)
        else:
            ty = ops.nc_lin2(tuple(tx_r.shape), ta, C1, x, C2, C0, flags=2, slope=slope, x2_period=n)
        ctx.save_for_backward(tx, x, gamma, mean, q, smean, ssd, ta, T1, T2, U[0], U[1])
        ctx.slope = slope
        return ty, tstyle

    @staticmethod
    def backward(ctx, h, u):
        if torch.is_grad_enabled():
            raise RuntimeError("DualTailBatchedFn is first-order only")
        tx, x, gamma, mean, q, smean, ssd, ta, T1, T2, U1, U2 = ctx.saved_tensors
        slope = ctx.slope
        n = x.shape[0]
        sp = _spatial(x)
        if h is None and u is None:
            return torch.zeros_like(tx), torch.zeros_like(x), torch.zeros_like(gamma), None, None, None, None, None
        if h is None:
            h = torch.zeros_like(tx[n:])
        if u is None:
            u = torch.zeros((n, 2 * x.shape[-1]), device=x.device, dtype=torch.float32)
        h = _cg(h)
        if ta is None or (FUSED_R1_TAIL and x.shape[-1] % 4 == 0):
            # two passes over the stacked cotangent instead of five: the three reductions in one, then both gradients in one
            lazy = ta is None                       # (the forward pass did not store ta: tx[n:] takes its place, times lrelu'(x) in the pass)
            t_op = tx[n:] if lazy else ta
            H1, H2, E = ops.nc_reduce_hxt(h, x, t_op, slope, ta_is_tx=lazy)
            co = ops.dual_tail_coef_bwd((H1, H2), E, _cg(u), (T1, T2), (U1, U2), mean, q, smean, ssd, gamma, sp)
            g_x, g_tx = ops.dual_tail_gx_tx(h, t_op, tx, x, co, slope, ta_is_tx=lazy)
            return g_tx, g_x, co["ggamma"], None, None, None, None, None
        H = ops.nc_reduce(h, x, flags=2, slope=slope, x2_period=n)          # sum h, sum h*lrelu(x)
        E = ops.nc_reduce(h, ta, want_sum=False)[1]                          # sum h*ta
        co = ops.dual_tail_coef_bwd(H, E, _cg(u), (T1, T2), (U1, U2), mean, q, smean, ssd, gamma, sp)
        g_tx = torch.empty_like(tx)
        ops.nc_lin2(tuple(ta.shape), h, co["K1"], x, co["K2"], co["K0"], flags=2 | 4, slope=slope, x2_period=n, out=g_tx[n:])
        ops.nc_lin2(tuple(x.shape), x, co["D2"], b=co["D0"], out=g_tx[:n])
        g_x = ops.dual_tail_gx(h, ta, tx, x, co, slope)
        return g_tx, g_x, co["ggamma"], None, None, None, None, None


def instance_norm(x, gamma, beta, eps=1e-3):
    """InstanceNormalization (instance_normalization.py:108-131): (x-mean)/(std+eps)*gamma+beta."""
    inv = 1.0 / _spatial(x)
    s1, s2 = nc_stats(x)
    mu = s1 * inv
    var = torch.clamp(s2 * inv - mu * mu, min=0.0)
    a = gamma.unsqueeze(0) / (torch.sqrt(var) + eps)
    return nc_lin(x, a, None, None, beta.unsqueeze(0) - mu * a)


def layer_style(x, eps=1e-6):
    """get_layer_style (confignet_utils.py:147-159) flattened as DiscrBlock does: (N, 2C) = [mu | std]."""
    inv = 1.0 / _spatial(x)
    s1, s2 = nc_stats(x)
    mu = s1 * inv
    var = torch.clamp(s2 * inv - mu * mu, min=0.0)
    return torch.cat([mu, torch.sqrt(var + eps)], dim=1)


def global_avg_pool(x):
    return nc_stats(x)[0] * (1.0 / _spatial(x))


class ChannelAffineActFn(Function):
    """y = relu?(x*a[c] + b[c] (+ res)): BatchNormalization in inference mode folded to a per-channel
    affine (keras ResNet50; R9), optional residual add and ReLU.  First-order only."""

    @staticmethod
    def forward(ctx, x, a, b, res, relu):
        x = _cg(x)
        res = _cg(res) if res is not None else None
        y = ops.nc_lin2(tuple(x.shape), x, _cg(a), res, None, _cg(b), flags=8 if relu else 0, per_channel=True)
        ctx.save_for_backward(x, a, y if relu else None)
        ctx.relu, ctx.has_res = relu, res is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        if torch.is_grad_enabled():
            raise RuntimeError("ChannelAffineActFn is first-order only")
        x, a, y = ctx.saved_tensors
        g = _cg(gy)
        if ctx.needs_input_grad[0] and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and x.shape[-1] % 4 == 0 and BN_BWD_FUSED:
            # one pass: ReLU backward, gx = a * g, sum g, sum g * x (and g itself for the residual branch)
            gx, gres, gb, ga = ops.bn_act_bwd(g, y if ctx.relu else x, x, a, ACT_RELU if ctx.relu else ACT_NONE, ctx.has_res)
            return gx, ga, gb, (gres if ctx.has_res else None), None
        if ctx.relu:
            g = ops.act_bwd(g, y, ACT_RELU)
        gx = ga = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.nc_lin2(tuple(x.shape), g, a, per_channel=True)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gb, ga = ops.nc_reduce(g, x, per_channel=True)
            gb, ga = gb.reshape(-1), ga.reshape(-1)
        return gx, ga, gb, (g if ctx.has_res else None), None


def channel_affine_act(x, a, b, res=None, relu=False):
    return ChannelAffineActFn.apply(x, a, b, res, relu)


# =============================================================================================
# pooling, pre-processing, rotation, loss reductions
# =============================================================================================
class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, s, pad):
        x = _cg(x)
        ctx.save_for_backward(x)
        ctx.cfg = (k, s, pad)
        return ops.maxpool_fwd(x, k, s, pad)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return ops.maxpool_bwd(x, _cg(gy), *ctx.cfg), None, None, None


def maxpool(x, k, s, pad=0):
    return MaxPoolFn.apply(x, k, s, pad)


class ChanAffine3Fn(Function):
    @staticmethod
    def forward(ctx, x, perm, scale, off):
        ctx.perm, ctx.scale = perm, scale
        return ops.chan_affine3_fwd(_cg(x), perm, scale, off)

    @staticmethod
    def backward(ctx, gy):
        return ops.chan_affine3_bwd(_cg(gy), ctx.perm, ctx.scale), None, None, None


def caffe_preprocess(x):
    """(x+1)*127.5, RGB<->BGR flip, subtract (103.939, 116.779, 123.68) (R8)."""
    return ChanAffine3Fn.apply(x, (2, 1, 0), 127.5, (127.5 - 103.939, 127.5 - 116.779, 127.5 - 123.68))


def vggface_preprocess(x):
    """(x+1)*127.5 - (93.5940, 104.7624, 129.1863), no flip (perceptual_loss.py:53-56)."""
    return ChanAffine3Fn.apply(x, (0, 1, 2), 127.5, (127.5 - 93.5940, 127.5 - 104.7624, 127.5 - 129.1863))


class Rotate3dFn(Function):
    @staticmethod
    def forward(ctx, grid, rot):
        grid, rot = _cg(grid), _cg(rot)
        ctx.save_for_backward(grid, rot)
        return ops.rotate3d_fwd(grid, rot)

    @staticmethod
    def backward(ctx, gout):
        grid, rot = ctx.saved_tensors
        ggrid, grot = ops.rotate3d_bwd(grid, rot, _cg(gout), ctx.needs_input_grad[1])
        return ggrid, grot


class EulerMatrixFn(Function):
    """euler_angles_to_matrix (confignet_utils.py:122-145) as one launch forward and one backward (cn_euler_matrix): as nine
    products of sines and cosines in torch it was ~110 launches per generator pass forward and ~150 in the backward pass."""

    @staticmethod
    def forward(ctx, angles):
        ctx.save_for_backward(angles)
        ctx.shape = angles.shape
        return ops.euler_matrix(angles)

    @staticmethod
    @torch.autograd.function.once_differentiable      # (a raw kernel: a double backward through the angles must raise, not treat
    def backward(ctx, g):                              # the Jacobian as constant -- euler_angles_to_matrix_composite is the twice
        (angles,) = ctx.saved_tensors                  # differentiable form)
        return ops.euler_matrix_bwd(angles, _cg(g)).reshape(ctx.shape)


def euler_angles_to_matrix(angles):
    return EulerMatrixFn.apply(angles)


def euler_angles_to_matrix_composite(angles):
    """The same matrix from torch operators (twice differentiable; kept as the cross-check of EulerMatrixFn)."""
    a = angles.reshape(-1, 3)
    s, c = torch.sin(a), torch.cos(a)
    rows = [c[:, 2] * c[:, 1], -s[:, 2], c[:, 2] * s[:, 1],
            s[:, 0] * s[:, 1] + c[:, 0] * c[:, 1] * s[:, 2], c[:, 0] * c[:, 2],
            c[:, 0] * s[:, 2] * s[:, 1] - c[:, 1] * s[:, 0],
            c[:, 1] * s[:, 0] * s[:, 2] - c[:, 0] * s[:, 1], c[:, 2] * s[:, 0],
            c[:, 0] * c[:, 1] + s[:, 0] * s[:, 1] * s[:, 2]]
    return torch.stack(rows, dim=-1).reshape(-1, 3, 3)


def rotate3d(grid, angles):
    return Rotate3dFn.apply(grid, euler_angles_to_matrix(angles))


class SqDiffSumFn(Function):
    """scale * sum (a-b)^2 ; gradient only to `a` (b is the ground-truth branch)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = _cg(a), _cg(b)
        ctx.save_for_backward(a, b)
        ctx.scale = scale
        return ops.sqdiff_sum(a, b, scale).reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ops.row_scale_diff(a, b, _cg(g.reshape(1)), 2.0 * ctx.scale), None, None


class SqDiffGroupSumFn(Function):
    """(G,) vector of mean((a - b)^2) over consecutive groups of samples (sizes along the batch axis); gradient only to `a`.
    One stacked VGG pass serves the synthetic and the real half of the generator step, whose perceptual terms stay separate
    entries of the loss dict: forward = one reduction per group over its (contiguous) slice, backward = ONE pass over the whole
    tensor with a per-sample scale (cn_row_scale_diff) -- no autograd slices of activation-sized tensors."""

    @staticmethod
    def forward(ctx, a, b, sizes):
        a, b = _cg(a), _cg(b)
        per = a.numel() // a.shape[0]
        outs, n0 = [], 0
        for sz in sizes:
            outs.append(ops.sqdiff_sum(a[n0:n0 + sz], b[n0:n0 + sz], 1.0 / (sz * per)))
            n0 += sz
        assert n0 == a.shape[0]
        ctx.save_for_backward(a, b)
        ctx.sizes, ctx.per = tuple(sizes), per
        return torch.cat(outs)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        s = torch.cat([g[i:i + 1].expand(sz) * (1.0 / (sz * ctx.per)) for i, sz in enumerate(ctx.sizes)]).contiguous()
        return ops.row_scale_diff(a, b, s, 2.0), None, None


def mse_group_sums(a, b, sizes):
    return SqDiffGroupSumFn.apply(a, b.detach(), tuple(int(s) for s in sizes))


def mse_sum(a, b):
    """mean((a-b)^2) over all elements (one perceptual-loss term, perceptual_loss.py:74-80)."""
    return SqDiffSumFn.apply(a, b.detach(), 1.0 / a.numel())


class RowSumSqFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _cg(x)
        ctx.save_for_backward(x)
        return ops.row_sumsq(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return RowScaleFn.apply(x, g, 2.0)


class RowScaleFn(Function):
    @staticmethod
    def forward(ctx, x, s, k):
        x, s = _cg(x), _cg(s)
        ctx.save_for_backward(x, s)
        ctx.k = k
        return ops.row_scale(x, s, k)

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        gx = RowScaleFn.apply(g, s, ctx.k) if ctx.needs_input_grad[0] else None
        gs = None
        if ctx.needs_input_grad[1]:
            gs = ctx.k * (ops.mul(_cg(g), x).reshape(x.shape[0], -1).sum(dim=1))
        return gx, gs, None


def row_sumsq(x):
    return RowSumSqFn.apply(x)


class MaskedDiffFn(Function):
    """(gt - gen) * mask; gradient to gen only (losses.py:14)."""

    @staticmethod
    def forward(ctx, gen, gt, mask):
        ctx.save_for_backward(mask)
        return ops.masked_diff(_cg(gt), _cg(gen), mask)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        gm = ops.masked_diff(_cg(g), None, mask)
        return ops.axpby(gm, None, -1.0, 0.0), None, None


class GanLossFn(Function):
    """mean(l*softplus(-s) + (1-l)*softplus(s)) for a constant label l (losses.py:7-11)."""

    @staticmethod
    def forward(ctx, scores, label):
        scores = _cg(scores)
        ctx.save_for_backward(scores)
        ctx.label = label
        return ops.gan_loss_fwd(scores, label).reshape(())

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        return ops.gan_loss_bwd(s, _cg(g.reshape(1)), ctx.label).reshape(s.shape), None


def gan_loss(scores, label):
    return GanLossFn.apply(scores, float(label))


class GanLossGroupFn(Function):
    """GanLossFn for every head of one discriminator call (losses.py:20-47: six score tensors per call): one launch forward, one
    backward (the node runs once the cotangents of all its scalars have arrived)."""

    @staticmethod
    def forward(ctx, labels, *scores):
        ctx.set_materialize_grads(False)
        scores = [_cg(s_) for s_ in scores]
        ctx.save_for_backward(*scores)
        ctx.labels = labels
        res = ops.gan_loss_grouped(scores, labels)
        return tuple(res[j].reshape(()) for j in range(len(scores)))

    @staticmethod
    def backward(ctx, *gs):
        scores = ctx.saved_tensors
        gouts = [None if g is None else _cg(g.reshape(1)).float() for g in gs]
        grads = ops.gan_loss_grouped(list(scores), ctx.labels, gouts, backward=True)
        return (None,) + tuple(g.reshape(s_.shape) for g, s_ in zip(grads, scores))


def gan_losses(scores, labels):
    """[mean(l softplus(-s) + (1 - l) softplus(s))] for the score tensors of several heads and their constant labels."""
    scores = list(scores)
    if len(scores) == 1 or any(s_.dtype != torch.float32 for s_ in scores):
        return [gan_loss(s_, l) for s_, l in zip(scores, labels)]
    return list(GanLossGroupFn.apply(tuple(float(l) for l in labels), *scores))
