"""Raw (non-differentiable) wrappers: torch CUDA tensors in, C-ABI call on the current HIP
stream, torch tensors out.  torch only allocates memory and provides the stream here."""
import contextlib
import ctypes
import math
import os
import threading

import torch

from ._lib import CN_BF16, CN_EUNSUPPORTED, CN_F32, CnConvGeom, CnDepthJob, CnGanJob, CnRowsJob, CnSumJob, check, lib

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3

# Storage type of ACTIVATION tensors (channels-last tensors with more than 4 channels).  fp32 is the reference's arithmetic
# (BASELINE.json configs[1]); bf16 is configs[2]: bf16 activations and bf16 filter copies on v_mfma_f32_32x32x16_bf16 with
# fp32 accumulation, while 3-channel images, (N, F) latent-sized tensors, statistics, coefficients, losses, master weights,
# their gradients and the optimizer state stay fp32.
ACT_DTYPE = torch.float32


def set_activation_dtype(dtype):
    """dtype: torch.float32 / torch.bfloat16 (or "f32" / "bf16").  Process-wide; switch only between steps."""
    global ACT_DTYPE
    if isinstance(dtype, str):
        dtype = {"f32": torch.float32, "fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                 "bfloat16": torch.bfloat16}[dtype]
    assert dtype in (torch.float32, torch.bfloat16)
    ACT_DTYPE = dtype


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Deterministic mode (CN_DETERMINISTIC=1 or set_deterministic(True); include/confignet_hip.h: cn_set_deterministic): every
# reduction of the fp32 path in a fixed order instead of fp32 atomics -- two runs on the same inputs are bit-identical.  Also
# switches off what reorders ADDS between streams: the generator step's fork (both branches add into the generator's slots).
DETERMINISTIC = False


def set_deterministic(on=True):
    global DETERMINISTIC
    check(lib.cn_set_deterministic(int(bool(on))), "cn_set_deterministic")
    DETERMINISTIC = bool(on)


def _partial_rows(rows):
    """Partial rows of a per-channel reduction over `rows` rows: every workgroup ends in one atomic per channel, so a long
    reduction is spread over partial rows (as if they were samples) that are added afterwards."""
    if rows < 8192:
        return 1
    return next(r for r in (16, 8, 4, 2, 1) if rows % r == 0)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.bfloat16, torch.uint8, torch.int64, torch.int32, torch.int16) and t.is_contiguous(), \
        "confignet_amd ops need contiguous CUDA tensors (got %s %s contiguous=%s)" % (t.device, t.dtype, t.is_contiguous())
    return ctypes.c_void_p(t.data_ptr())


def _fptr(t):
    """Pointer of a tensor the ABI declares as float*."""
    assert t is None or t.dtype == torch.float32, "fp32 tensor expected, got %s" % (t.dtype,)
    return _ptr(t)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _dt(t):
    return CN_BF16 if t.dtype == torch.bfloat16 else CN_F32


def cast(x, dtype):
    """Storage-type conversion fp32 <-> bf16 (round to nearest even) by a HIP kernel."""
    if x.dtype == dtype:
        return x
    x = _c(x)
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    check(lib.cn_cast(_ptr(x), _dt(x), _ptr(out), _dt(out), x.numel(), _stream()), "cn_cast")
    return out


def f32(x):
    return x if x is None else cast(x, torch.float32)


def _unify(*ts):
    """Activation operands of one elementwise call share a storage type: bf16 as soon as one of them is (an fp32 gradient
    arriving from a dense layer meets a bf16 activation)."""
    dt = torch.bfloat16 if any(t is not None and t.dtype == torch.bfloat16 for t in ts) else torch.float32
    return [None if t is None else cast(t, dt) for t in ts]


def _act_out_dtype(channels):
    return ACT_DTYPE if channels > 4 else torch.float32


# ---------------------------------------------------------------------------------------------
# convolution geometry ([TF-2.1] SAME rule R2: total = max((ceil(e/s)-1)*s + k - e, 0), lo = total//2)
# ---------------------------------------------------------------------------------------------
class ConvSpec:
    """Static description of one conv layer call: spatial kernel, stride, folded x2 upsample,
    optional explicit symmetric padding (ZeroPadding2D + 'valid')."""

    def __init__(self, kernel, stride=1, up=0, explicit_pad=None):
        self.kernel = tuple(kernel)
        self.stride, self.up, self.explicit_pad = stride, up, explicit_pad

    def geom(self, x_shape, cout):
        nd = len(self.kernel)
        assert len(x_shape) == nd + 2
        n, cin = x_shape[0], x_shape[-1]
        sp = list(x_shape[1:-1])
        ins, outs, pads = [1, 1, 1], [1, 1, 1], [0, 0, 0]
        ks = [1, 1, 1]
        for i in range(nd):
            a = 3 - nd + i
            e = sp[i] << self.up
            k, s = self.kernel[i], self.stride
            if self.explicit_pad is None:
                o = -(-e // s)
                total = max((o - 1) * s + k - e, 0)
                lo = total // 2
            else:
                lo = self.explicit_pad
                o = (e + 2 * lo - k) // s + 1
            ins[a], outs[a], pads[a], ks[a] = sp[i], o, lo, k
        g = CnConvGeom()
        g.nd, g.n = nd, n
        g.in_d, g.in_h, g.in_w, g.cin = ins[0], ins[1], ins[2], cin
        g.out_d, g.out_h, g.out_w, g.cout = outs[0], outs[1], outs[2], cout
        g.k_d, g.k_h, g.k_w = ks
        s = self.stride
        g.s_d, g.s_h, g.s_w = (s if nd == 3 else 1), s, s
        g.dl_d = g.dl_h = g.dl_w = 1
        g.p_d, g.p_h, g.p_w = pads
        g.up = self.up
        return g


def geom_out_shape(g):
    return (g.n, g.out_d, g.out_h, g.out_w, g.cout) if g.nd == 3 else (g.n, g.out_h, g.out_w, g.cout)


def geom_in_shape(g, upsampled=False):
    u = g.up if upsampled else 0
    if g.nd == 3:
        return (g.n, g.in_d << u, g.in_h << u, g.in_w << u, g.cin)
    return (g.n, g.in_h << u, g.in_w << u, g.cin)


_CACHE_SINGLE = False      # A/B: one derived copy per filter, re-derived whenever the other branch used it
_keepalive = None        # a list while an InferenceGraph is captured: every derived filter copy the capture references


def _weight_cache(w, slot, make):
    val = _weight_cache_lookup(w, slot, make)
    if _keepalive is not None:
        _keepalive.append(val)
    return val


def _weight_cache_lookup(w, slot, make):
    """Derived forms of a filter (tap-flipped fp32 copy, bf16 operand copies), cached on the tensor.  A weight of a network
    (nn.Net tags its tensors with their owner) is re-derived when THAT network's epoch moves; other tensors (the cotangent
    "filters" of double-backward calls) when torch's version counter does.  The key also holds the stream (a forked step
    derives on each branch instead of sharing a tensor across streams) and the global generation, which a HIP-graph capture
    bumps so that copies used inside a graph are made inside it -- except for FROZEN networks (no trainable weight: the VGG
    stacks), whose copies are made once, outside any capture, and then referenced by every graph."""
    from .nn import WEIGHTS_EPOCH
    owner = getattr(w, "_cn_owner", None)
    frozen = owner is not None and getattr(owner, "n_trainable", 1) == 0
    if frozen:
        key = (owner.epoch, w.data_ptr())
        c = getattr(w, slot, None)
        if c is not None and c[0] == key:
            return c[1]
        if not torch.cuda.is_current_stream_capturing():
            val = make(w.detach())
            torch.cuda.current_stream().synchronize()   # once per frozen filter: any stream may read the copy from now on
            setattr(w, slot, (key, val))
            return val
        return make(w.detach())                    # first use happens inside a capture: a graph-private copy
    stream = torch.cuda.current_stream().cuda_stream if w.is_cuda else 0
    key = (WEIGHTS_EPOCH[0], owner.epoch if owner is not None else -1, w._version, w.data_ptr(), stream)
    c = getattr(w, slot, None)
    # one entry PER STREAM (a dict): the two branches of a forked step use the same generator / regressor filters alternately --
    # forward on one stream, forward on the other, then the two backward passes -- and a single entry made each of them derive its
    # copy again (the key holds the stream, so the other branch's entry never matched and was replaced)
    if isinstance(c, dict) and not _CACHE_SINGLE:
        e = c.get(stream)
        if e is not None and e[0] == key:
            return e[1]
    val = make(w.detach())
    try:
        if not isinstance(c, dict) or any(e[0][:4] != key[:4] for e in c.values()):
            c = {}                                   # another epoch / version / storage: drop every stream's stale copy
        c[stream] = (key, val)
        setattr(w, slot, c)
    except Exception:
        pass
    return val


def weight_prep_bf16(w):
    """(wf [t][co][ci], wd [t][ci][co]) bf16 operand copies of the fp32 filter w [..taps.., cin, cout]."""
    def make(wd_):
        taps, cin, cout = int(math.prod(wd_.shape[:-2])), wd_.shape[-2], wd_.shape[-1]
        both = torch.empty((2, taps * cin * cout), device=wd_.device, dtype=torch.bfloat16)
        check(lib.cn_conv_weight_prep_bf16(_fptr(_c(wd_)), _ptr(both[0]), _ptr(both[1]), taps, cin, cout, _stream()),
              "cn_conv_weight_prep_bf16")
        return both[0], both[1]
    return _weight_cache(w, "_cn_wbf16", make)


def _bf16_conv_ok(g):
    return ACT_DTYPE == torch.bfloat16 and g.cin % 8 == 0 and g.cout % 8 == 0


# ---------------------------------------------------------------------------------------------
# x2-upsample-folded convolutions collapsed per output-parity class (include/confignet_hip.h: cn_upfold_*)
# ---------------------------------------------------------------------------------------------
UPFOLD = True


def upfold_ok(g):
    """Generator layers UpSampling + Conv(k3 / k4, SAME, stride 1) with wide channels (map_final keeps its own kernel)."""
    return (UPFOLD and g.up == 1 and g.cin > 4 and g.cout > 4 and g.s_d * g.s_h * g.s_w == 1 and g.dl_d * g.dl_h * g.dl_w == 1
            and all(2 <= k <= 4 for k in ((g.k_d, g.k_h, g.k_w) if g.nd == 3 else (g.k_h, g.k_w)))
            and g.out_h == 2 * g.in_h and g.out_w == 2 * g.in_w and (g.nd == 2 or g.out_d == 2 * g.in_d))


def _copy_geom(g, **kw):
    out = CnConvGeom()
    for name, _ in CnConvGeom._fields_:
        setattr(out, name, getattr(g, name))
    for k, v in kw.items():
        setattr(out, k, v)
    return out


def upfold_prepare(w, g):
    """(wf, wd, gd, g2): class filters (fp32, cached on the filter per weight epoch) and the two geometries they are used
    with (rebuilt per call: batch and extents belong to the call, not to the filter)."""
    def make(wd_):
        k3 = (ctypes.c_int * 3)(g.k_d, g.k_h, g.k_w)
        p3 = (ctypes.c_int * 3)(g.p_d, g.p_h, g.p_w)
        k2, p2 = (ctypes.c_int * 3)(), (ctypes.c_int * 3)()
        wd_ = _c(wd_)
        ks = [g.k_d, g.k_h, g.k_w]
        sizes = [1 if (g.nd == 2 and i == 0) else (ks[i] + 1 if ks[i] >= 3 else 3) for i in range(3)]   # k2 per axis
        t2 = sizes[0] * sizes[1] * sizes[2]
        both = torch.empty((2, t2 * g.cin * g.cout), device=wd_.device, dtype=torch.float32)
        check(lib.cn_upfold_weights(_fptr(wd_), _ptr(both[0]), _ptr(both[1]), g.nd, k3, p3, g.cin, g.cout, k2, p2, _stream()),
              "cn_upfold_weights")
        assert list(k2) == sizes, (list(k2), sizes)
        return both[0], both[1], tuple(k2), tuple(p2)
    wf, wd, (kd, kh, kw), (pd, ph, pw) = _weight_cache(w, "_cn_upfold_%d_%d%d%d_%d%d%d" % (g.nd, g.k_d, g.k_h, g.k_w, g.p_d, g.p_h, g.p_w), make)
    # forward: zero-stuffed (dl = 2) convolution on the stored grid, parity-ordered rows, dead taps skipped
    gd = _copy_geom(g, k_d=kd, k_h=kh, k_w=kw, dl_d=(2 if g.nd == 3 else 1), dl_h=2, dl_w=2,
                    p_d=kd - 1 - pd, p_h=kh - 1 - ph, p_w=kw - 1 - pw, up=0)
    # conv2: from the output grid (cout channels) to the stored grid (cin channels), stride 2
    g2 = _copy_geom(g, in_d=g.out_d, in_h=g.out_h, in_w=g.out_w, cin=g.cout, out_d=g.in_d, out_h=g.in_h, out_w=g.in_w,
                    cout=g.cin, k_d=kd, k_h=kh, k_w=kw, s_d=(2 if g.nd == 3 else 1), s_h=2, s_w=2, p_d=pd, p_h=ph, p_w=pw, up=0)
    taps = (kd, kh, kw) if g.nd == 3 else (kh, kw)
    return wf.view(taps + (g.cin, g.cout)), wd.view(taps + (g.cout, g.cin)), gd, g2


def upfold_wgrad(gw2, g, w_shape, out=None, single_writer=False):
    """gw[kk][ci][co] from the conv2 filter gradient gw2[a][co][ci]; out: add to this tensor instead (atomically, unless the
    caller is the only one adding to it right now)."""
    gw = out if out is not None else torch.empty(w_shape, device=gw2.device, dtype=torch.float32)
    k3 = (ctypes.c_int * 3)(g.k_d, g.k_h, g.k_w)
    p3 = (ctypes.c_int * 3)(g.p_d, g.p_h, g.p_w)
    mode = 0 if out is None else (2 if single_writer else 1)
    check(lib.cn_upfold_wgrad(_fptr(_c(gw2)), _ptr(gw), g.nd, k3, p3, g.cin, g.cout, mode, _stream()), "cn_upfold_wgrad")
    return gw


# ---------------------------------------------------------------------------------------------
# Winograd F(2x2, 3x3) for the wide 2-D 3x3 stride-1 SAME layers (fp32 mode): include/confignet_hip.h: cn_conv_fwd_wino
# ---------------------------------------------------------------------------------------------
WINOGRAD = True
WINO_MIN_WGS = 32           # (round 6, alternating pipelined runs: 128 -> 64 -> 32 = 409.6 / 411.3 / 411.5 images/s; what counts next to other lines is the work issued, not the isolated latency)
WINO_MIN_FILL = 0.7         # fraction of the 8 x 8-tile blocks that must lie inside the image


def _wino_ok(g, cin, cout):
    """cin / cout: reduction / output channels of the convolution actually run (swapped for a data gradient)."""
    if not (WINOGRAD and g.nd == 2 and g.k_h == 3 and g.k_w == 3 and g.s_h == 1 and g.s_w == 1 and g.dl_h == 1 and g.dl_w == 1
            and g.up == 0 and g.p_h == 1 and g.p_w == 1 and g.out_h == g.in_h and g.out_w == g.in_w):
        return False
    if cin % 16 or cout % 64:
        return False
    # workgroup = 8 x 8 tiles (16 x 16 output pixels) x 64 channels: most of the block must be image, and there must be
    # enough workgroups to fill the chip (below that the direct kernel with split-K wins)
    th, tw = (g.in_h + 1) // 2, (g.in_w + 1) // 2
    bh, bw = (th + 7) // 8, (tw + 7) // 8
    if th * tw < WINO_MIN_FILL * (bh * bw * 64):
        return False
    return g.n * bh * bw * (cout // 64) >= WINO_MIN_WGS


WINO4 = True
WINO4_MIN_WGS = 128


def _wino4_ok(g, cin, cout):
    """Winograd F(4x4, 3x3) (csrc/winograd4.hip) takes the 3x3 stride-1 SAME layers whose image divides into its 16 x 32-pixel
    blocks and that fill the chip with them (VGG-19 conv1_2 .. conv3_4 at the benchmark's size); cin / cout as in _wino_ok."""
    if not (WINO4 and WINOGRAD and g.nd == 2 and g.k_h == 3 and g.k_w == 3 and g.s_h == 1 and g.s_w == 1 and g.dl_h == 1 and g.dl_w == 1
            and g.up == 0 and g.p_h == 1 and g.p_w == 1 and g.out_h == g.in_h and g.out_w == g.in_w):
        return False
    if g.in_h % 16 or g.in_w % 32 or cin % 16 or cout % 64:
        return False
    return g.n * (g.in_h // 16) * (g.in_w // 32) * (cout // 64) >= WINO4_MIN_WGS


def _wino4_filter(w, dgrad):
    def make(wd_):
        wd_ = _c(wd_)
        cin, cout = wd_.shape[-2], wd_.shape[-1]
        u = torch.empty((36, cout, cin) if dgrad else (36, cin, cout), device=wd_.device, dtype=torch.float32)
        check(lib.cn_conv_wino4_filter(_fptr(wd_), _ptr(u), cin, cout, int(dgrad), _stream()), "cn_conv_wino4_filter")
        return u
    return _weight_cache(w, "_cn_wino4_d" if dgrad else "_cn_wino4_f", make)


def wino4_saved_flops(g):
    """Direct-convolution multiply-adds (border taps counted) minus the 36 per 4x4 tile that F(4x4, 3x3) issues."""
    tiles = g.n * (g.in_h // 4) * (g.in_w // 4)
    return 2.0 * g.cin * g.cout * (9.0 * g.n * g.in_h * g.in_w - 36.0 * tiles)


def _wino_filter(w, dgrad):
    def make(wd_):
        wd_ = _c(wd_)
        cin, cout = wd_.shape[-2], wd_.shape[-1]
        u = torch.empty((16, cout, cin) if dgrad else (16, cin, cout), device=wd_.device, dtype=torch.float32)
        check(lib.cn_conv_wino_filter(_fptr(wd_), _ptr(u), cin, cout, int(dgrad), _stream()), "cn_conv_wino_filter")
        return u
    return _weight_cache(w, "_cn_wino_d" if dgrad else "_cn_wino_f", make)


# Statistics of a convolution's output taken in its epilogue (cn_conv_fwd_stats) for the normalisation layer that follows:
# `with request_stats(kind[, slope])` around the convolution, `take_stats(y, kind)` in the consumer.  kind "act": (sum a, sum a^2) of
# the stored, activated output (AdaIn); kind "pre4": (sum v, sum v^2, sum l, sum l^2), l = leaky_relu(v, slope), of a convolution
# without activation (DiscrBlock tail).  Only launches that can carry them do (fp32, the unsplit LDS-DMA loop, tiles inside one
# sample, a zero pool active, not the deterministic mode); everything else leaves no entry and the consumer runs its own pass.
class _StatsSlot(threading.local):        # per host thread: a request / a result never crosses to a convolution issued by another thread
    request = None
    ready = None                          # (data_ptr, shape, kind, tensors) of the most recent fused launch


_stats = _StatsSlot()
STATS_FUSION = True


class request_stats:
    def __init__(self, kind, slope=0.0):
        self.req = (kind, float(slope))

    def __enter__(self):
        self.prev, _stats.request = _stats.request, (self.req if STATS_FUSION and not DETERMINISTIC else None)

    def __exit__(self, *exc):
        _stats.request = self.prev


def take_stats(y, kind):
    """The statistics the producing convolution left for tensor y, or None."""
    ent, _stats.ready = _stats.ready, None
    if ent is not None and ent[0] == y.data_ptr() and ent[1] == tuple(y.shape) and ent[2] == kind:
        return ent[3]
    return None


def conv_fwd(x, w, bias, g, act=ACT_NONE, slope=0.0):
    req, _stats.request = _stats.request, None           # (a request applies to the next convolution only)
    _stats.ready = None
    out_dtype = _act_out_dtype(g.cout)
    if ACT_DTYPE == torch.float32 and x.dtype == torch.float32 and _wino4_ok(g, g.cin, g.cout):
        y = torch.empty(geom_out_shape(g), device=x.device, dtype=torch.float32)
        check(lib.cn_conv_fwd_wino4(g.n, g.in_h, g.in_w, g.cin, g.cout, _ptr(_c(x)), _ptr(_wino4_filter(w, False)), _fptr(bias), _ptr(y),
                                    act, slope, _stream()), "cn_conv_fwd_wino4")
        prof_note_saved(wino4_saved_flops(g))
        return y
    if ACT_DTYPE == torch.float32 and x.dtype == torch.float32 and _wino_ok(g, g.cin, g.cout):
        y = torch.empty(geom_out_shape(g), device=x.device, dtype=torch.float32)
        check(lib.cn_conv_fwd_wino(g.n, g.in_h, g.in_w, g.cin, g.cout, _ptr(x), _ptr(_wino_filter(w, False)), _fptr(bias), _ptr(y),
                                   act, slope, _stream()), "cn_conv_fwd_wino")
        prof_note_saved(wino_saved_flops(g))
        return y
    if _bf16_conv_ok(g):
        x = cast(x, torch.bfloat16)
        y = torch.empty(geom_out_shape(g), device=x.device, dtype=torch.bfloat16)
        wf, _ = weight_prep_bf16(w)
        check(lib.cn_conv_fwd_bf16(ctypes.byref(g), _ptr(x), _ptr(wf), _fptr(bias), _ptr(y), act, slope, _stream()), "cn_conv_fwd_bf16")
        return y
    x = f32(x)                                  # 3-channel image layers (and everything in fp32 mode): fp32 MFMA family
    if out_dtype == torch.bfloat16 and MIXED_FIRST_LAYERS and g.cin == 3:
        # first layer of the bf16 path: fp32 image in, bf16 out of the same kernel (no conversion pass over the wide tensor)
        y = torch.empty(geom_out_shape(g), device=x.device, dtype=torch.bfloat16)
        rc = lib.cn_conv_fwd_dt(ctypes.byref(g), _ptr(x), CN_F32, _fptr(w), _fptr(bias), _ptr(y), CN_BF16, act, slope, _stream())
        if rc == 0:
            return y
        if rc != CN_EUNSUPPORTED:
            check(rc, "cn_conv_fwd_dt")
    y = torch.empty(geom_out_shape(g), device=x.device, dtype=torch.float32)
    if req is not None and out_dtype == torch.float32 and g.cout % 4 == 0 and (req[0] == "act" or act == ACT_NONE):
        nk = 2 if req[0] == "act" else 4
        mark = _active_pool.cur if _active_pool is not None else None
        st = zero_pool_alloc((nk, g.n, g.cout), x.device)
        if st is not None:
            rc = lib.cn_conv_fwd_stats(ctypes.byref(g), _ptr(_c(x)), _fptr(w), _fptr(bias), _ptr(y), act, slope, _ptr(st), 1 if nk == 2 else 2,
                                       req[1], _stream())
            if rc == 0:
                _stats.ready = (y.data_ptr(), tuple(y.shape), req[0], tuple(st[k] for k in range(nk)))
                return y
            if rc != CN_EUNSUPPORTED:
                check(rc, "cn_conv_fwd_stats")
            _active_pool.cur = mark                   # nothing was launched: hand the (untouched, still zero) slab back to the pool
    check(lib.cn_conv_fwd(ctypes.byref(g), _ptr(x), _fptr(w), _fptr(bias), _ptr(y), act, slope, _stream()), "cn_conv_fwd")
    return cast(y, out_dtype)


def conv_fwd_res(x, w, bias, res, g, act=ACT_NONE, slope=0.0):
    """act(conv(x, w) + bias + res): the residual add of a ResNet block in the convolution's epilogue where the launch carries it
    (cn_conv_fwd_res: unsplit implicit-GEMM launches), else convolution + one nc_lin2 pass."""
    assert act in (ACT_NONE, ACT_RELU)
    if ACT_DTYPE == torch.float32 and x.dtype == torch.float32 and res.dtype == torch.float32 and not _wino_ok(g, g.cin, g.cout):
        x, res = _c(x), _c(res)
        y = torch.empty(geom_out_shape(g), device=x.device, dtype=torch.float32)
        rc = lib.cn_conv_fwd_res(ctypes.byref(g), _ptr(x), _fptr(w), _fptr(bias), _ptr(res), _ptr(y), act, slope, _stream())
        if rc == 0:
            return y
        if rc != CN_EUNSUPPORTED:
            check(rc, "cn_conv_fwd_res")
    y = conv_fwd(x, w, bias, g)
    return nc_lin2(tuple(y.shape), y, None, res, None, None, flags=8 if act == ACT_RELU else 0)


def scale_columns_segments(src, seg, a, total):
    """Packed copy of the filters listed in `seg` (int32 (nseg, 5) on the device: see cn_scale_columns_segments), each scaled per
    output channel by its slice of `a`."""
    dst = torch.empty(total, device=src.device, dtype=torch.float32)
    check(lib.cn_scale_columns_segments(_fptr(src), _ptr(dst), _ptr(seg), _fptr(_c(a)), seg.shape[0], total, _stream()),
          "cn_scale_columns_segments")
    return dst


def weight_tflip(w):
    w = _c(w)
    taps = int(math.prod(w.shape[:-2]))
    wt = torch.empty(w.shape[:-2] + (w.shape[-1], w.shape[-2]), device=w.device, dtype=torch.float32)
    check(lib.cn_conv_weight_tflip(_fptr(w), _ptr(wt), taps, w.shape[-2], w.shape[-1], _stream()), "cn_conv_weight_tflip")
    return wt


def conv_dgrad(gy, w, g):
    """Gradient w.r.t. the (virtually upsampled) input of the conv described by g; w is the filter itself (the operand
    copies the kernels want are derived and cached here)."""
    shape = geom_in_shape(g, upsampled=True)
    if _bf16_conv_ok(g):
        gy = cast(gy, torch.bfloat16)
        gu = torch.empty(shape, device=gy.device, dtype=torch.bfloat16)
        _, wd = weight_prep_bf16(w)
        check(lib.cn_conv_dgrad_bf16(ctypes.byref(g), _ptr(gy), _ptr(wd), _ptr(gu), _stream()), "cn_conv_dgrad_bf16")
        return gu
    if gy.dtype == torch.bfloat16 and MIXED_FIRST_LAYERS and g.cin == 3:
        # data gradient into the fp32 image straight from the bf16 output gradient
        gu = torch.empty(shape, device=gy.device, dtype=torch.float32)
        wt = _weight_cache(w, "_cn_tflip", weight_tflip)
        rc = lib.cn_conv_dgrad_dt(ctypes.byref(g), _ptr(_c(gy)), CN_BF16, _fptr(wt), _ptr(gu), CN_F32, _stream())
        if rc == 0:
            return cast(gu, _act_out_dtype(g.cin))
        if rc != CN_EUNSUPPORTED:
            check(rc, "cn_conv_dgrad_dt")
    gy = f32(gy)
    gu = torch.empty(shape, device=gy.device, dtype=torch.float32)
    if ACT_DTYPE == torch.float32 and _wino4_ok(g, g.cout, g.cin):
        check(lib.cn_conv_fwd_wino4(g.n, g.in_h, g.in_w, g.cout, g.cin, _ptr(_c(gy)), _ptr(_wino4_filter(w, True)), None, _ptr(gu),
                                    ACT_NONE, 0.0, _stream()), "cn_conv_fwd_wino4")
        prof_note_saved(wino4_saved_flops(g))
        return gu
    if ACT_DTYPE == torch.float32 and _wino_ok(g, g.cout, g.cin):
        check(lib.cn_conv_fwd_wino(g.n, g.in_h, g.in_w, g.cout, g.cin, _ptr(gy), _ptr(_wino_filter(w, True)), None, _ptr(gu),
                                   ACT_NONE, 0.0, _stream()), "cn_conv_fwd_wino")
        prof_note_saved(wino_saved_flops(g))
        return gu
    if DGRAD_FROM_W:
        # the kernel transposes the filter tile on its way into LDS: no tap-flipped copy per trainable filter per step
        rc = lib.cn_conv_dgrad_w(ctypes.byref(g), _ptr(gy), _fptr(_c(w)), _ptr(gu), _stream())
        if rc == 0:
            return cast(gu, _act_out_dtype(g.cin))
        if rc != CN_EUNSUPPORTED:
            check(rc, "cn_conv_dgrad_w")
    wt = _weight_cache(w, "_cn_tflip", weight_tflip)
    check(lib.cn_conv_dgrad(ctypes.byref(g), _ptr(gy), _fptr(wt), _ptr(gu), _stream()), "cn_conv_dgrad")
    return cast(gu, _act_out_dtype(g.cin))


def conv_dgrad_res(gy, w, g, res):
    """conv_dgrad(gy, w, g) + res: the gradient of a tensor that feeds this convolution and a skip connection, the add in the
    data-gradient launch's epilogue where the launch carries it (cn_conv_dgrad_w_res: fp32, stride 1, unsplit implicit-GEMM
    launches), else data gradient + one nc_lin2 pass."""
    if (ACT_DTYPE == torch.float32 and gy.dtype == torch.float32 and res.dtype == torch.float32 and DGRAD_FROM_W and not g.up
            and not _wino4_ok(g, g.cout, g.cin) and not _wino_ok(g, g.cout, g.cin)):
        gu = torch.empty(geom_in_shape(g, upsampled=True), device=gy.device, dtype=torch.float32)
        rc = lib.cn_conv_dgrad_w_res(ctypes.byref(g), _ptr(_c(gy)), _fptr(_c(w)), _ptr(_c(res)), _ptr(gu), _stream())
        if rc == 0:
            return gu
        if rc != CN_EUNSUPPORTED:
            check(rc, "cn_conv_dgrad_w_res")
    gu = conv_dgrad(gy, w, g)
    return nc_lin2(tuple(gu.shape), gu, None, res, None, None)


DGRAD_FROM_W = True
THIN_WGRAD = True
C3_WGRAD = True
MIXED_FIRST_LAYERS = True      # bf16 path: first-layer kernels that read / write both storage types
_C3_PARTS = []


def _c3_partials():
    if not _C3_PARTS:
        _C3_PARTS.append(int(lib.cn_conv_wgrad_c3_partials()))
    return _C3_PARTS[0]


_EXP_NO_WGRAD = False       # timing experiment only (wrong gradients): filter gradients not launched


def conv_wgrad(x, gy, g, w_shape, out=None, accumulate=True):
    """out: ADD the filter gradient to this tensor (a slot of a gradient arena, see grad_sink) instead of returning a new one;
    out with accumulate=False: WRITE it there (an uninitialised scratch the caller owns: the folded ResNet-50's packed gradients)."""
    if _EXP_NO_WGRAD:
        return out if out is not None else torch.empty(w_shape, device=x.device, dtype=torch.float32)
    gw = out if out is not None else zero_pool_alloc(w_shape, x.device)
    pre = gw is not None and (accumulate or out is None)
    if gw is None:
        gw = torch.empty(w_shape, device=x.device, dtype=torch.float32)
    if C3_WGRAD and g.nd == 2 and g.cin == 3 and g.k_h == 3 and g.k_w == 3 and g.s_h == g.s_w and g.s_h in (1, 2) \
            and g.dl_h == 1 and g.dl_w == 1 and g.up == 0 and g.cout <= 64 and g.cout % 4 == 0 and gy.dtype in (torch.float32, torch.bfloat16):
        # K = 27 first layers: staged-tile kernel without atomics (the generic split-over-rows kernel runs them at 10 TFLOP/s)
        x, gy = _c(f32(x)), _c(gy)
        scratch = torch.empty(_c3_partials() * 27 * g.cout, device=x.device, dtype=torch.float32)
        check(lib.cn_conv_wgrad_c3(ctypes.byref(g), _ptr(x), _ptr(gy), _dt(gy), _ptr(scratch), _fptr(gw), int(out is not None and accumulate), _stream()),
              "cn_conv_wgrad_c3")
        return gw
    if THIN_WGRAD and g.nd == 2 and g.cout <= 4 and 8 <= g.cin <= 64 and g.cin % 4 == 0 and g.s_h == 1 and g.s_w == 1:
        # thin output (map_final): staged-tile VALU kernel with ordered partial sums (the implicit-GEMM kernel runs it at 6 TFLOP/s)
        x32, gy32 = _c(f32(x)), _c(f32(gy))
        taps = g.k_h * g.k_w
        scratch = torch.empty(int(lib.cn_conv_wgrad_thin_partials()) * taps * g.cin * g.cout, device=x.device, dtype=torch.float32)
        rc = lib.cn_conv_wgrad_thin(ctypes.byref(g), _ptr(x32), _ptr(gy32), _ptr(scratch), _fptr(gw), int(pre), _stream())
        if rc == 0:
            return gw
        if rc != CN_EUNSUPPORTED:
            check(rc, "cn_conv_wgrad_thin")
    if _bf16_conv_ok(g):
        x, gy = cast(x, torch.bfloat16), cast(gy, torch.bfloat16)
        check(lib.cn_conv_wgrad_bf16(ctypes.byref(g), _ptr(x), _ptr(gy), _fptr(gw), int(pre), _stream()), "cn_conv_wgrad_bf16")
        return gw
    x, gy = f32(x), f32(gy)
    # caller-owned workspace for the partial filters of the kernel's row splits (0 bytes: a single split / the fall-back kernel)
    nbytes = int(lib.cn_conv_wgrad_workspace_bytes(ctypes.byref(g)))
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32) if nbytes else None
    if nbytes and out is not None and _SINK is not None and DEFER_SLAB_SUMS:
        # inside a backward pass (grad_sink): the row slices' slabs stay in `ws` and the pass adds the slabs of ALL its filter
        # gradients with one grouped launch at its join (grad_sink.join) instead of one reduction launch per layer on the chain
        parts = ctypes.c_int(0)
        check(lib.cn_conv_wgrad_ws_slabs(ctypes.byref(g), _ptr(x), _ptr(gy), _fptr(gw), int(pre), _ptr(ws), nbytes, ctypes.byref(parts),
                                         _stream()), "cn_conv_wgrad_ws_slabs")
        if parts.value:
            _SINK.setdefault("slabs", []).append((ws, gw, parts.value, gw.numel(), int(pre)))
        return gw
    check(lib.cn_conv_wgrad_ws(ctypes.byref(g), _ptr(x), _ptr(gy), _fptr(gw), int(pre), _ptr(ws), nbytes, _stream()), "cn_conv_wgrad_ws")
    return gw


# ---------------------------------------------------------------------------------------------
# gradient sink: while nn.backward_into_arenas runs a backward pass, the filter / dense-weight / bias gradients are ADDED by
# their kernels straight into the weights' slots of the networks' gradient arenas (tf.GradientTape sums the contributions of
# every use of a variable: here the wgrad kernels' own accumulate mode does, not autograd's add kernels and not a copy pass
# afterwards).  GRAD_SINK = False: gradients through autograd as before.
# Nothing downstream of a weight gradient is on the backward chain, so the sinks CAN run on a side stream that forks off the
# chain (WGRAD_FORK = True, one cross-stream edge per WGRAD_GROUP launches, joined at the end of the pass; all sinks share
# ONE side stream, so accumulations into one slot never race).  Measured and left OFF: with the filter gradients not
# launched at all the iteration drops 49.8 -> 42.3 ms (the bound of the idea), but the forked graphs replay SLOWER than the
# single chain -- 63.6 ms with an edge per 8 launches, 53.2 per 32, 52.3 with one fork at the end of the pass, against
# 49.5 unforked: the four hardware queues already carry the iteration's four concurrent lines, a fifth branch takes a queue
# from one of them (GPU_MAX_HW_QUEUES = 6 / 8 make every variant worse: 61 - 87 ms).
# ---------------------------------------------------------------------------------------------
GRAD_SINK = True
DEFER_SLAB_SUMS = True     # the filter gradients' slab reductions of a backward pass as ONE grouped launch at its join (False: one per layer)
WGRAD_FORK = False
# Round 4 experiment, OFF (WGRAD_BALANCE = True switches it on): load balancing between the TWO streams a forked step already has.
# The backward pass of the generator step runs as two chains -- the real branch (VGG, generator, the whole ResNet-50 encoder) on
# the model's branch stream, the synthetic branch on the calling stream -- and the real chain is the longer one by the encoder's
# backward.  Filter / dense-weight gradients are leaves of the tape (nothing on a chain waits for them), so the sink launches
# issued on a stream other than the one the pass was started on were queued and handed to the calling stream in groups (one
# cross-stream edge per WGRAD_GROUP launches; no extra stream, no extra hardware queue -- what sank WGRAD_FORK).  Measured
# (profiles/round4_schedule_experiments.txt): 335 / 327 / 334 images/s with an edge per 8 / 4 / 16 launches against 352 without --
# every cross-branch edge inside a captured graph costs more than the idle tail of the shorter chain it would fill.
WGRAD_BALANCE = False
_SINK = None              # {"slots": {data_ptr: grad view}, "side": stream | None, "keep": [...], "used": bool}
_SIDE_STREAMS = {}


class grad_sink:
    def __init__(self, params):
        self.params = params

    def __enter__(self):
        global _SINK
        assert _SINK is None, "nested gradient sinks"
        if not GRAD_SINK or not self.params:
            return self
        dev = self.params[0].device
        side = None
        if WGRAD_FORK:
            side = _SIDE_STREAMS.get(dev)
            if side is None:
                side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
        _SINK = {"slots": {p.data_ptr(): p.grad for p in self.params if p.grad is not None}, "side": side, "keep": [], "used": False,
                 "home": torch.cuda.current_stream(dev)}
        return self

    def join(self):
        """Order the calling stream after everything the sinks launched; drop the tensors kept alive for them."""
        st = _SINK
        _sink_flush()
        _balance_flush()
        if st is not None and st["side"] is not None and st["used"]:
            torch.cuda.current_stream().wait_stream(st["side"])
        if st is not None:
            cur = torch.cuda.current_stream()
            touched = st.pop("touched", [])
            for s_ in touched:                          # every stream a sink launched on (a forked step has two) before anything
                if s_ != cur:                           # below reads or adds to the arenas on the calling stream
                    cur.wait_stream(s_)
            slabs = st.pop("slabs", [])
            if slabs:
                # every filter gradient of the pass that split its rows: (slabs, destination, parts, count, accumulate) -> ONE launch
                # per 80 of them, each reduced in slice order exactly as cn_conv_wgrad_ws would have (same bits)
                # (a weight used more than once in the pass -- the generator runs twice in the generator step -- has several jobs
                # with ONE destination: they go into successive launches, in the order of their filter gradients, so that the adds
                # into one element keep a fixed order; typically two or three launches for ~70 reductions)
                rounds, seen = [], {}
                for job in slabs:
                    r = seen.get(job[1].data_ptr(), 0)
                    seen[job[1].data_ptr()] = r + 1
                    if r == len(rounds):
                        rounds.append([])
                    rounds[r].append(job)
                for batch in rounds:
                    jobs = (CnSumJob * len(batch))()
                    for q, (ws, dst, parts, count, acc) in zip(jobs, batch):
                        q.src, q.dst, q.count, q.parts, q.accumulate = ws.data_ptr(), dst.data_ptr(), count, parts, acc
                    check(lib.cn_sum_parts_grouped(jobs, len(batch), _stream()), "cn_sum_parts_grouped")
            depth = st.pop("depth", [])
            if depth:
                rounds, seen = [], {}                 # (same-destination jobs in successive launches, as above)
                for job in depth:
                    r = seen.get(job[2].data_ptr(), 0)
                    seen[job[2].data_ptr()] = r + 1
                    if r == len(rounds):
                        rounds.append([])
                    rounds[r].append(job)
                for batch in rounds:
                    jobs = (CnDepthJob * len(batch))()
                    for q, (a, b, slot) in zip(jobs, batch):
                        q.a, q.b, q.c = a.data_ptr(), b.data_ptr(), slot.data_ptr()
                        q.m, q.n, q.k, q.lda, q.ldb, q.ldc = a.shape[1], b.shape[1], a.shape[0], a.shape[1], b.shape[1], b.shape[1]
                    check(lib.cn_gemm_depth_grouped(jobs, len(batch), _stream()), "cn_gemm_depth_grouped")
            if slabs or depth:
                if not torch.cuda.is_current_stream_capturing():      # (eager dispatch: the operands were allocated on the stream of
                    for t in [j[0] for j in slabs] + [t for j in depth for t in j[:2]]:     # their producer, not on this one)
                        t.record_stream(cur)
                for s_ in touched:                    # a stream that produced operands must not run ahead of these launches: a block
                    if s_ != cur:                     # the allocator hands out again after the join would be overwritten under them
                        s_.wait_stream(cur)
            for gw2, g, w_shape, slot in st.pop("upfold", {}).values():
                upfold_wgrad(gw2, g, w_shape, out=slot, single_writer=True)
            post = st.pop("post", [])
            for fn in post:                           # launches that read what the grouped reductions above have just completed
                fn()                                  # (the folded ResNet-50's parameter gradients: cn_bn_fold_bwd)
            post_keep = st.pop("post_keep", [])
            if post:
                if not torch.cuda.is_current_stream_capturing():
                    for t in post_keep:
                        t.record_stream(cur)
                for s_ in touched:
                    if s_ != cur:
                        s_.wait_stream(cur)
        if st is not None:
            st["keep"].clear()
            st["used"] = False

    def __exit__(self, exc_type, exc, tb):
        global _SINK
        try:
            if exc_type is None:
                self.join()
            elif _SINK is not None:                 # the pass raised: no scatter work on half-written scratch, just let go
                _SINK["keep"].clear()
        finally:
            _SINK = None                            # a failed pass must not leave every later one dying on "nested gradient sinks"


def sink_for(w):
    """The gradient-arena slot of weight w if a sink is active for it (else None)."""
    if _SINK is None or w is None:
        return None
    return _SINK["slots"].get(w.data_ptr())


WGRAD_GROUP = 8      # sink launches per fork off the backward chain


def _sink_flush():
    """Fork the side stream off the calling stream HERE and launch everything queued since the last fork on it."""
    st = _SINK
    if st is None or not st.get("queue"):
        return
    queue, st["queue"] = st["queue"], []
    side = st["side"]
    for s_ in st.pop("streams", ()):          # every stream a queued launch's operands were produced on (a forked step has two)
        side.wait_stream(s_)
    st["used"] = True
    with torch.cuda.stream(side):
        for fn in queue:
            fn()


def _balance_flush():
    """WGRAD_BALANCE: launch the sinks queued from branch streams on the stream the backward pass was started on, ordered after
    what their operands' streams have been issued so far."""
    st = _SINK
    if st is None or not st.get("bq"):
        return
    queue, st["bq"] = st["bq"], []
    home = st["home"]
    for s_ in st.pop("bq_streams", ()):
        home.wait_stream(s_)
    with torch.cuda.stream(home):
        for fn in queue:
            fn()


def _sink_run(fn, keep):
    """Run one sink launch: immediately on the calling stream (no fork), or queued for the side stream -- every WGRAD_GROUP
    launches one cross-stream edge (an edge per layer made the captured graph a ladder that replayed 35 % slower than the
    single chain: 49.3 -> 66.5 ms).  `keep`: the tensors the launch reads; they stay alive until the join."""
    st = _SINK
    if st is not None:
        cur0 = torch.cuda.current_stream()
        if all(cur0 != s_ for s_ in st.setdefault("touched", [])):
            st["touched"].append(cur0)
    if st is not None and st["side"] is None and WGRAD_BALANCE and not DETERMINISTIC and cur0 != st["home"]:
        capturing = torch.cuda.is_current_stream_capturing()
        for t in keep:
            if t is not None:
                st["keep"].append(t)
                if not capturing:
                    t.record_stream(st["home"])       # (eager dispatch: the block was allocated on the branch stream)
        st.setdefault("bq", []).append(fn)
        if all(cur0 != s_ for s_ in st.setdefault("bq_streams", [])):
            st["bq_streams"].append(cur0)
        if len(st["bq"]) >= WGRAD_GROUP:
            _balance_flush()
        return
    if st is None or st["side"] is None:
        fn()
        return
    st["keep"].extend(t for t in keep if t is not None)
    st.setdefault("queue", []).append(fn)
    cur = torch.cuda.current_stream()
    if all(cur != s_ for s_ in st.setdefault("streams", [])):
        st["streams"].append(cur)
    if len(st["queue"]) >= WGRAD_GROUP:
        _sink_flush()


def sink_conv_wgrad(x, gy, g, w_shape, slot):
    _sink_run(lambda: conv_wgrad(x, gy, g, w_shape, out=slot), (x, gy))


def sink_conv_wgrad_to(x, gy, g, w_shape, dst):
    """The filter gradient WRITTEN (not added) into the caller's scratch `dst`, launched like a sink (a leaf of the pass)."""
    _sink_run(lambda: conv_wgrad(x, gy, g, w_shape, out=dst, accumulate=False), (x, gy))


def sink_post(fn, keep=()):
    """Run fn() once every sink launch of the pass -- the deferred grouped reductions included -- has been issued: at the join of
    the active gradient sink, on the stream that joins; immediately without one."""
    st = _SINK
    if st is None:
        fn()
        return
    cur0 = torch.cuda.current_stream()
    if all(cur0 != s_ for s_ in st.setdefault("touched", [])):
        st["touched"].append(cur0)
    st.setdefault("post_keep", []).extend(t for t in keep if t is not None)
    st.setdefault("post", []).append(fn)


def bn_fold_bwd(seg9, blocks, gwf, gshift, arena, a, rs, bm, gout):
    """cn_bn_fold_bwd: gout (laid out like `arena`) += the kernel / bias / gamma / beta gradients of every folded conv + BN pair."""
    check(lib.cn_bn_fold_bwd(_ptr(seg9), seg9.shape[0], blocks, _fptr(gwf), _fptr(gshift), _fptr(arena), _fptr(a), _fptr(rs), _fptr(bm),
                             _fptr(gout), _stream()), "cn_bn_fold_bwd")


def sink_upfold_wgrad(gy, x, g2, wd_shape, g, w_shape, slot):
    """Upsample-folded layer: the class-filter gradient of EVERY use of the weight in this backward pass (the generator runs twice
    in the generator step) is accumulated into one scratch by the filter-gradient kernel (atomics: any stream), and scattered
    back to the k taps ONCE, at the join of the pass, by a single writer (10 atomic scatters of up to 172 us -> 5 plain ones)."""
    st = _SINK
    pend = st.setdefault("upfold", {})
    ent = pend.get(slot.data_ptr())
    if ent is None:
        gw2 = zero_pool_alloc(wd_shape, x.device)
        if gw2 is None:
            gw2 = torch.zeros(wd_shape, device=x.device, dtype=torch.float32)
        ent = pend[slot.data_ptr()] = (gw2, g, w_shape, slot)
    gw2 = ent[0]
    _sink_run(lambda: conv_wgrad(gy, x, g2, wd_shape, out=gw2), (x, gy))


def sink_gemm(a, b, slot, trans_a=False, trans_b=False):
    st = _SINK
    if (st is not None and DEFER_SLAB_SUMS and trans_a and not trans_b and a.shape[0] <= 32 and a.dtype == torch.float32
            and b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous() and slot.is_contiguous()):
        # a dense layer's weight gradient x^T gy over a batch of <= 32 rows: a leaf of the pass -- queued for ONE grouped launch at
        # the pass' join (grad_sink.join -> cn_gemm_depth_grouped) instead of a launch of its own on the backward chain
        cur0 = torch.cuda.current_stream()
        if all(cur0 != s_ for s_ in st.setdefault("touched", [])):
            st["touched"].append(cur0)
        st.setdefault("depth", []).append((a, b, slot))
        return
    _sink_run(lambda: gemm_acc(a, b, slot, trans_a, trans_b), (a, b))


# ---------------------------------------------------------------------------------------------
# per-step zero pool: accumulate-into outputs (statistics sums, filter gradients) are carved from a buffer that
# is cleared by ONE launch at the start of a training step instead of one clearing launch per call
# ---------------------------------------------------------------------------------------------
class _ZeroPool:
    def __init__(self, device, numel):
        self.buf = torch.zeros(numel, device=device, dtype=torch.float32)
        self.cur = 0
        self.dirty = 0          # high-water mark of the previous use


_pools = {}
_active_pool = None
ZERO_POOL_FLOATS = 96 * 1024 * 1024


def zero_pool_begin(name, device):
    """Activate pool `name` (created on first use) and clear what its previous use dirtied."""
    global _active_pool
    p = _pools.get((name, device))
    if p is None:
        p = _pools[(name, device)] = _ZeroPool(device, ZERO_POOL_FLOATS)
    if p.dirty:
        p.buf[:p.dirty].zero_()
    p.cur = 0
    _active_pool = p


def zero_pool_end():
    global _active_pool
    if _active_pool is not None:
        _active_pool.dirty = max(_active_pool.dirty, _active_pool.cur)
    _active_pool = None


def zero_pool_alloc(shape, device):
    p = _active_pool
    if p is None or p.buf.device != device:
        return None
    n = int(math.prod(shape))
    n4 = (n + 3) // 4 * 4
    if p.cur + n4 > p.buf.numel():
        return None
    out = p.buf[p.cur:p.cur + n].view(shape)
    p.cur += n4
    return out


def zero_(t):
    """Clear a contiguous fp32 tensor with the library's own kernel (a graph node that re-executes on replay)."""
    check(lib.cn_zero(_fptr(t), t.numel() * 4, _stream()), "cn_zero")
    return t


def sumpool2(gu):
    nd = gu.dim() - 2
    sp = [s // 2 for s in gu.shape[1:-1]]
    gx = torch.empty((gu.shape[0], *sp, gu.shape[-1]), device=gu.device, dtype=gu.dtype)
    d, h, w = ([1] + sp) if nd == 2 else sp
    check(lib.cn_sumpool2(_ptr(gu), _ptr(gx), nd, gu.shape[0], d, h, w, gu.shape[-1], _dt(gu), _stream()), "cn_sumpool2")
    return gx


# ---------------------------------------------------------------------------------------------
def gemm(a, b, trans_a=False, trans_b=False, bias=None, act=ACT_NONE, slope=0.0):
    """C = act(op(a) @ op(b) + bias); a, b 2-D row-major."""
    m = a.shape[1] if trans_a else a.shape[0]
    k = a.shape[0] if trans_a else a.shape[1]
    n = b.shape[0] if trans_b else b.shape[1]
    kb = b.shape[1] if trans_b else b.shape[0]
    assert k == kb, "gemm inner dims %d vs %d" % (k, kb)
    a, b = f32(a), f32(b)                      # dense layers compute in fp32 (a flattened bf16 feature map is converted)
    c = torch.empty((m, n), device=a.device, dtype=torch.float32)
    check(lib.cn_gemm(int(trans_a), int(trans_b), m, n, k, _ptr(a), a.shape[1], _ptr(b), b.shape[1], _ptr(c), n,
                      _ptr(bias), act, slope, _stream()), "cn_gemm")
    return c


def gemm_acc(a, b, out, trans_a=False, trans_b=False):
    """out += op(a) @ op(b) (cn_gemm_acc)."""
    m = a.shape[1] if trans_a else a.shape[0]
    k = a.shape[0] if trans_a else a.shape[1]
    n = b.shape[0] if trans_b else b.shape[1]
    a, b = f32(a), f32(b)
    assert tuple(out.shape) == (m, n) and out.is_contiguous()
    check(lib.cn_gemm_acc(int(trans_a), int(trans_b), m, n, k, _ptr(a), a.shape[1], _ptr(b), b.shape[1], _fptr(out), n, _stream()),
          "cn_gemm_acc")
    return out


def _nsc(x):
    n, c = x.shape[0], x.shape[-1]
    return n, x.numel() // (n * c), c


def nc_reduce(x1, x2=None, want_sum=True, want_dot=True, flags=0, slope=0.0, per_channel=False, x2_period=0):
    """(sum_s f1(x1), sum_s f1(x1)*f2(x2 or x1)) per (n, c); per_channel folds n into s.
    x2_period: x2 holds only that many samples and sample n of x1 pairs with x2[n % x2_period]."""
    x1, x2 = _unify(x1, x2)
    n, s, c = _nsc(x1)
    assert not (x2_period and per_channel)
    flags |= x2_period << 8
    rep = 1
    if per_channel:
        # every workgroup ends in one atomic per channel: spread a long reduction over `rep` partial rows (as if
        # they were samples) so that no address takes more than ~128 of them, then add the partials
        rows = n * s
        rep = _partial_rows(rows)
        n, s = rep, rows // rep
    if want_sum and want_dot:           # adjacent outputs: cleared by one launch (or by the step's zero pool)
        s12 = zero_pool_alloc((2, n, c), x1.device)
        if s12 is not None:
            flags |= 16
        else:
            s12 = torch.empty((2, n, c), device=x1.device, dtype=torch.float32)
        s1, s2 = s12[0], s12[1]
    else:
        one = zero_pool_alloc((n, c), x1.device)
        if one is not None:
            flags |= 16
        else:
            one = torch.empty((n, c), device=x1.device, dtype=torch.float32)
        s1, s2 = (one, None) if want_sum else (None, one)
    check(lib.cn_nc_reduce(_ptr(x1), _ptr(x2), _ptr(s1), _ptr(s2), n, s, c, flags, slope, _dt(x1), _stream()), "cn_nc_reduce")
    if rep > 1:
        if want_sum and want_dot:
            s12 = s12.sum(1, keepdim=True)
            return s12[0], s12[1]
        s1 = s1.sum(0, keepdim=True) if s1 is not None else None
        s2 = s2.sum(0, keepdim=True) if s2 is not None else None
    return s1, s2


def nc_reduce4(x, slope):
    """(sum x, sum x^2, sum l, sum l^2) per (n, c) with l = leaky_relu(x, slope): cn_nc_reduce4 (one pass instead of two)."""
    n, s, c = _nsc(x)
    out = zero_pool_alloc((4, n, c), x.device)
    flags = 16
    if out is None:
        out, flags = torch.empty((4, n, c), device=x.device, dtype=torch.float32), 0
    check(lib.cn_nc_reduce4(_ptr(x), _ptr(out), n, s, c, slope, flags, _dt(x), _stream()), "cn_nc_reduce4")
    return out[0], out[1], out[2], out[3]


def nc_lin2(shape, x1=None, a1=None, x2=None, a2=None, b=None, flags=0, slope=0.0, per_channel=False, a3=None, b3=None,
            x2_period=0, out=None):
    """y = a1*f1(x1) + a2*f2(x2) + b with (n,c) [or (c,)] coefficients broadcast over space.
    x2_period: as in nc_reduce.  out: write into this (contiguous) tensor instead of a new one."""
    n, c = shape[0], shape[-1]
    s = int(math.prod(shape)) // (n * c)
    x1, x2 = _unify(x1, x2)
    if flags & 4:
        _log_mask(x2)                              # the result is multiplied by lrelu'(x2)
    ref = x1 if x1 is not None else (x2 if x2 is not None else b)
    dtype = ref.dtype if (x1 is not None or x2 is not None) else _act_out_dtype(c)
    if out is not None:
        assert out.is_contiguous() and out.numel() == int(math.prod(shape))
        if out.dtype != dtype:                     # the caller's buffer decides the storage type
            x1, x2 = (None if x1 is None else cast(x1, out.dtype)), (None if x2 is None else cast(x2, out.dtype))
        y = out
    else:
        y = torch.empty(shape, device=ref.device, dtype=dtype)
    flags |= x2_period << 8
    check(lib.cn_nc_lin2(_ptr(x1), _fptr(a1), _ptr(x2), _fptr(a2), _fptr(b), _fptr(a3), _fptr(b3), _ptr(y), n, s, c,
                         0 if per_channel else c, flags, slope, _dt(y), _stream()), "cn_nc_lin2")
    return y


NORM_ADAIN, NORM_INSTANCE, NORM_STYLE = 0, 1, 2


def norm_coef_fwd(mode, s1, s2, p1, p2, spatial, eps):
    n, c = s1.shape
    A = torch.empty((n, 2 * c) if mode == NORM_STYLE else (n, c), device=s1.device, dtype=torch.float32)
    B = torch.empty((n, c), device=s1.device, dtype=torch.float32) if mode != NORM_STYLE else None
    mean = torch.empty((n, c), device=s1.device, dtype=torch.float32)
    r = torch.empty((n, c), device=s1.device, dtype=torch.float32)
    check(lib.cn_norm_coef_fwd(mode, _ptr(s1), _ptr(s2), _ptr(p1), _ptr(p2), _ptr(A), _ptr(B), _ptr(mean), _ptr(r),
                               n, c, spatial, eps, _stream()), "cn_norm_coef_fwd")
    return A, B, mean, r


def norm_coef_bwd(mode, t1, t2, mean, r, p1, spatial, eps):
    n, c = mean.shape
    dev = mean.device
    c1 = torch.empty((n, c), device=dev, dtype=torch.float32) if mode != NORM_STYLE else None
    c2 = torch.empty((n, c), device=dev, dtype=torch.float32)
    c0 = torch.empty((n, c), device=dev, dtype=torch.float32)
    gp1 = gp2 = None
    if mode == NORM_ADAIN:
        gp1 = torch.empty((n, 2 * c), device=dev, dtype=torch.float32)
    elif mode == NORM_INSTANCE:
        gp1 = torch.empty((c,), device=dev, dtype=torch.float32)
        gp2 = torch.empty((c,), device=dev, dtype=torch.float32)
    check(lib.cn_norm_coef_bwd(mode, _ptr(t1), _ptr(t2), _ptr(mean), _ptr(r), _ptr(p1), _ptr(c1), _ptr(c2), _ptr(c0),
                               _ptr(gp1), _ptr(gp2), n, c, spatial, eps, _stream()), "cn_norm_coef_bwd")
    return c1, c2, c0, gp1, gp2


# AdaIn / instance norm: the coefficient algebra inline in the apply / backward pass (cn_norm_apply) where it fits.  -110 launches per
# iteration; on the iteration's time it is neutral for fp32 storage (425.8 against 426.1 images/s, alternating runs) and was -0.3 %
# for bf16 storage (762.4 against 764.8: every workgroup of the short bf16 pass starts with the dependent loads of the statistics),
# so bf16 tensors keep the separate coefficient launch.
NORM_APPLY = True


def norm_apply_fwd(mode, x, s1, s2, p1, p2, eps, flags=0, slope=0.0):
    """(y, mean, r) = the apply pass of AdaIn / instance norm with its coefficients computed inline (cn_norm_apply, dir 0), or None
    where the launch does not fit (the caller then runs norm_coef_fwd + nc_lin2)."""
    if not NORM_APPLY or x.shape[-1] % 4 or x.dtype != torch.float32:
        return None
    n, s, c = _nsc(x)
    y = torch.empty_like(x)
    mean = torch.empty((n, c), device=x.device, dtype=torch.float32)
    r = torch.empty_like(mean)
    rc = lib.cn_norm_apply(mode, 0, _ptr(x), None, _fptr(s1), _fptr(s2), _fptr(_c(p1)), _fptr(None if p2 is None else _c(p2)), _ptr(mean), _ptr(r),
                           None, None, None, None, _ptr(y), n, s, c, eps, flags, slope, _dt(x), _stream())
    if rc == CN_EUNSUPPORTED:
        return None
    check(rc, "cn_norm_apply")
    return y, mean, r


def norm_apply_bwd(mode, gy, x, t1, t2, mean, r, p1, eps, flags=0, slope=0.0, a3=None, b3=None):
    """(gx, gp1, gp2) = the input gradient of AdaIn / instance norm with the coefficients inline (cn_norm_apply, dir 1) and the
    parameter gradients (mode 0: gp1 = d[s|b] (n, 2c); mode 1: d gamma, d beta), or None where the launch does not fit."""
    if not NORM_APPLY or x.shape[-1] % 4 or x.dtype != torch.float32 or gy.dtype != torch.float32:
        return None
    gy, x = _unify(gy, x)
    n, s, c = _nsc(x)
    if flags & 4:
        _log_mask(x)
    gx = torch.empty_like(x)
    if mode == NORM_ADAIN:
        gp1, gp2 = torch.empty((n, 2 * c), device=x.device, dtype=torch.float32), None
    else:
        gp1 = torch.empty((c,), device=x.device, dtype=torch.float32)
        gp2 = torch.empty_like(gp1)
    rc = lib.cn_norm_apply(mode, 1, _ptr(gy), _ptr(x), _fptr(t1), _fptr(t2), _fptr(_c(p1)), None, _ptr(mean), _ptr(r), _ptr(gp1), _ptr(gp2),
                           _fptr(a3), _fptr(b3), _ptr(gx), n, s, c, eps, flags, slope, _dt(x), _stream())
    if rc == CN_EUNSUPPORTED:
        return None
    check(rc, "cn_norm_apply")
    return gx, gp1, gp2


def dual_tail_coef_fwd(T, U, mean, q, sm, ssd, gamma, spatial, eps=1e-3):
    """T = (T1, T2) or None (no ty wanted); U = (U1, U2) or None (no tstyle wanted).  One head at a time: the rows of T / U
    are the samples of the primal statistics.  Batched (both given): U holds the N samples of the head that leaves through the
    style statistics, T the (heads - 1) * N samples that go on, against N samples of primal statistics."""
    ref = mean if mean is not None else sm
    period, c = ref.shape
    mk = lambda *sh: torch.empty(sh, device=ref.device, dtype=torch.float32)
    C1 = C2 = C0 = ts = None
    n_inst = n_style = 0
    if T is not None:
        n_inst = T[0].shape[0]
        C1, C2, C0 = mk(n_inst, c), mk(n_inst, c), mk(n_inst, c)
    if U is not None:
        n_style = U[0].shape[0]
        ts = mk(n_style, 2 * c)
    T1, T2 = T if T is not None else (None, None)
    U1, U2 = U if U is not None else (None, None)
    check(lib.cn_dual_tail_coef_fwd(_ptr(T1), _ptr(T2), _ptr(U1), _ptr(U2), _ptr(mean), _ptr(q), _ptr(sm), _ptr(ssd),
                                    _ptr(gamma), _ptr(C1), _ptr(C2), _ptr(C0), _ptr(ts), n_inst + n_style, c, spatial, eps,
                                    n_style, period, _stream()),
          "cn_dual_tail_coef_fwd")
    return C1, C2, C0, ts


def dual_tail_coef_bwd(H, E, u, T, U, mean, q, sm, ssd, gamma, spatial, eps=1e-3):
    """Returns dict of coefficient tensors (see include/confignet_hip.h).  Row counts as in dual_tail_coef_fwd: the
    instance-norm group follows H (sum h ...), the style group follows u."""
    ref = mean if mean is not None else sm
    period, c = ref.shape
    names = ["K1", "K2", "K0", "D2", "D0", "kh", "kt", "ka", "kc", "et", "ex", "e0", "ggamma", "ggamma_rows"]
    out = {k: None for k in names}
    mk = lambda *sh: torch.empty(sh, device=ref.device, dtype=torch.float32)
    n_inst = n_style = 0
    if H is not None:
        n_inst = H[0].shape[0]
        for k in ("K1", "K2", "K0", "kh", "kt", "ka", "kc", "ggamma_rows"):
            out[k] = mk(n_inst, c)
        out["ggamma"] = mk(c)
    if u is not None:
        n_style = u.shape[0]
        for k in ("D2", "D0", "et", "ex", "e0"):
            out[k] = mk(n_style, c)
    arr = (ctypes.c_void_p * 14)(*[(out[k].data_ptr() if out[k] is not None else None) for k in names])
    H1, H2p = H if H is not None else (None, None)
    T1, T2 = T if T is not None else (None, None)
    U1, U2 = U if U is not None else (None, None)
    check(lib.cn_dual_tail_coef_bwd(_ptr(H1), _ptr(H2p), _ptr(E), _ptr(u), _ptr(T1), _ptr(T2), _ptr(U1), _ptr(U2),
                                    _ptr(mean), _ptr(q), _ptr(sm), _ptr(ssd), _ptr(gamma), arr, n_inst + n_style, c, spatial, eps,
                                    n_style, period, _stream()),
          "cn_dual_tail_coef_bwd")
    return out


def dual_tail_gx(h, ta, tx, x, co, slope):
    """h / ta may hold a multiple of x's samples (batched tangent pass: the heads that share this activation, summed here)."""
    h, ta, tx, x = _unify(h, ta, tx, x)
    _log_mask(x)
    n, s, c = _nsc(x)
    nrep = 1 if h is None else h.shape[0] // n
    assert h is None or (h.shape[0] == nrep * n and ta.shape[0] == nrep * n)
    out = torch.empty_like(x)
    check(lib.cn_dual_tail_gx(_ptr(h), _ptr(ta), _ptr(tx), _ptr(x), _ptr(co["kh"]), _ptr(co["kt"]), _ptr(co["ka"]),
                              _ptr(co["kc"]), _ptr(co["et"]), _ptr(co["ex"]), _ptr(co["e0"]), _ptr(out), n, s, c, slope,
                              nrep, _dt(x), _stream()), "cn_dual_tail_gx")
    return out


def dual_tail_gx_tx(h, ta, tx, x, co, slope, ta_is_tx=False):
    """(g_x, g_tx): dual_tail_gx and, from the same pass, the gradient w.r.t. the stacked tangent input tx (its first N samples
    the style head's D2 x + D0, then per head lrelu'(x) (K1 h + K2 lrelu(x) + K0)): cn_dual_tail_gx_tx."""
    h, ta, tx, x = _unify(h, ta, tx, x)
    _log_mask(x)
    n, s, c = _nsc(x)
    nrep = h.shape[0] // n
    assert h.shape[0] == nrep * n and ta.shape[0] == nrep * n and tx.shape[0] == (nrep + 1) * n
    out = torch.empty_like(x)
    out_tx = torch.empty_like(tx)
    check(lib.cn_dual_tail_gx_tx(_ptr(h), _ptr(ta), _ptr(tx), _ptr(x), _ptr(co["kh"]), _ptr(co["kt"]), _ptr(co["ka"]), _ptr(co["kc"]),
                                 _ptr(co["et"]), _ptr(co["ex"]), _ptr(co["e0"]), _ptr(co["K1"]), _ptr(co["K2"]), _ptr(co["K0"]),
                                 _ptr(co["D2"]), _ptr(co["D0"]), _ptr(out), _ptr(out_tx), n, s, c, slope, nrep, int(ta_is_tx), _dt(x), _stream()),
          "cn_dual_tail_gx_tx")
    return out, out_tx


def nc_reduce_hxt(h, x, ta, slope, ta_is_tx=False):
    """(sum h, sum h lrelu(x), sum h ta) per (n, c) of h in ONE pass (cn_nc_reduce_hxt); x holds h.shape[0] / k samples.
    ta_is_tx: `ta` is the tangent input tx, ta = lrelu'(x) tx is formed in the pass."""
    h, x, ta = _unify(h, x, ta)
    n, s, c = _nsc(h)
    out = zero_pool_alloc((3, n, c), h.device)
    flags = 16 | (32 if ta_is_tx else 0)
    if out is None:
        out, flags = torch.empty((3, n, c), device=h.device, dtype=torch.float32), flags & 32
    check(lib.cn_nc_reduce_hxt(_ptr(h), _ptr(x), _ptr(ta), _ptr(out), n, s, c, slope, x.shape[0], flags, _dt(h), _stream()), "cn_nc_reduce_hxt")
    return out[0], out[1], out[2]


def bn_act_bwd(gy, y, x, a, act, want_g):
    """(gx, g | None, sum_c g, sum_c g*x): cn_bn_act_bwd -- activation backward, BatchNorm(inference) input gradient and the two
    per-channel parameter sums in one pass over gy / y / x."""
    gy, y, x = _unify(gy, y, x)
    _log_mask(y, act)
    c = gy.shape[-1]
    rows = gy.numel() // c
    rep = _partial_rows(rows)
    gx = torch.empty_like(gy)
    g = torch.empty_like(gy) if want_g else None
    s12 = zero_pool_alloc((2, rep, c), gy.device)
    flags = 16
    if s12 is None:
        s12, flags = torch.empty((2, rep, c), device=gy.device, dtype=torch.float32), 0
    check(lib.cn_bn_act_bwd(_ptr(gy), _ptr(y), _ptr(x), _fptr(_c(a)), _ptr(g), _ptr(gx), _ptr(s12[0]), _ptr(s12[1]), rep, rows // rep, c,
                            act, flags, _dt(gy), _stream()), "cn_bn_act_bwd")
    if rep > 1:
        s12 = s12.sum(1)
    else:
        s12 = s12.reshape(2, c)
    return gx, g, s12[0], s12[1]


def nc_reduce_dact(x1, x2, act, slope, x2_period=0, flags=0, want_dot=True, want_a=True):
    """(a, sum_s a, sum_s a * f2(x2)) with a = x1 * act'(x2): cn_nc_reduce_dact (one pass instead of act_bwd + nc_reduce).
    want_a = False: the sums only (a is not stored: None)."""
    x1, x2 = _unify(x1, x2)
    _log_mask(x2, act)
    n, s, c = _nsc(x1)
    a = torch.empty_like(x1) if want_a else None
    flags |= x2_period << 8
    s12 = zero_pool_alloc((2, n, c), x1.device)
    if s12 is not None:
        flags |= 16
    else:
        s12 = torch.empty((2, n, c), device=x1.device, dtype=torch.float32)
    check(lib.cn_nc_reduce_dact(_ptr(x1), _ptr(x2), _ptr(s12[0]), _ptr(s12[1]) if want_dot else None, _ptr(a), n, s, c, flags, slope,
                                act, _dt(x1), _stream()), "cn_nc_reduce_dact")
    return a, s12[0], (s12[1] if want_dot else None)


# ---------------------------------------------------------------------------------------------
# Test instrument (tests/test_nets_gpu.py: whole-step gradient parity): with `branch_log()` active every backward-pass site that
# evaluates a piecewise-linear derivative -- LeakyReLU / ReLU masks, max-pool arg-max -- appends the tensor its decisions are
# taken from (as a host-side boolean mask `t > 0`, or the pool's fp32 input) to the list.  The float64 oracle then FORCES the
# same decisions (oracle.ref_ops.BranchControl(forced=...)), so product and oracle gradients differ by summation error only.
# Off (None) in the product path: one `is not None` test per call.
# ---------------------------------------------------------------------------------------------
BRANCH_LOG = None


class branch_log:
    def __enter__(self):
        global BRANCH_LOG
        BRANCH_LOG = self.log = []
        return self.log

    def __exit__(self, *exc):
        global BRANCH_LOG
        BRANCH_LOG = None


def _log_mask(t, act=None):
    if BRANCH_LOG is not None and (act is None or act in (ACT_LRELU, ACT_RELU)):
        BRANCH_LOG.append(("act", (t.detach().float() > 0).reshape(-1).cpu()))


def act_fwd(x, act, slope=0.0):
    y = torch.empty_like(x)
    check(lib.cn_act_fwd(_ptr(x), _ptr(y), x.numel(), act, slope, _dt(x), _stream()), "cn_act_fwd")
    return y


def act_bwd(gy, y, act, slope=0.0):
    gy, y = _unify(gy, y)
    _log_mask(y, act)
    gx = torch.empty_like(gy)
    check(lib.cn_act_bwd(_ptr(gy), _ptr(y), _ptr(gx), gy.numel(), act, slope, _dt(gy), _stream()), "cn_act_bwd")
    return gx


def act_bwd_partials(gy, y, act, slope=0.0):
    """(gx, partial (rep, c)): act_bwd and the per-channel sums of its result over `rep` row slices, in one pass
    (cn_act_bwd_bias); the caller adds the slices (sum_rows_into)."""
    gy, y = _unify(gy, y)
    _log_mask(y, act)
    _, _, c = _nsc(gy)
    rows = gy.numel() // c
    rep = _partial_rows(rows)                       # as nc_reduce(per_channel=True)
    gx = torch.empty_like(gy)
    gb = zero_pool_alloc((rep, c), gy.device)
    flags = 16
    if gb is None:
        gb, flags = torch.empty((rep, c), device=gy.device, dtype=torch.float32), 0
    check(lib.cn_act_bwd_bias(_ptr(gy), _ptr(y), _ptr(gx), _ptr(gb), rep, rows // rep, c, act, slope, flags, _dt(gy), _stream()),
          "cn_act_bwd_bias")
    return gx, gb


def act_bwd_bias(gy, y, act, slope=0.0, sink=None):
    """(gx, gb): act_bwd fused with the per-channel sum of its result (the bias gradient) -- one pass instead of two.
    sink: the bias's slot of a gradient arena (grad_sink) -- the sum is ADDED there and gb is returned as None."""
    gx, gb = act_bwd_partials(gy, y, act, slope)
    rep = gb.shape[0]
    if sink is not None:
        sum_rows_into(gb, sink)
        return gx, None
    return gx, (gb.sum(0) if rep > 1 else gb.reshape(-1))


def sum_rows_into(partial, dst, accumulate=True, side=True):
    """dst[c] (+)= sum_r partial[r][c] (cn_sum_rows_into); under a grad_sink on its side stream (off the backward chain)."""
    rows, c = partial.shape
    if side and accumulate and _SINK is not None and DEFER_SLAB_SUMS and partial.dtype == torch.float32 and partial.is_contiguous():
        # a leaf of the backward pass: joins the pass' grouped reduction launch (grad_sink.join) -- dst[c] += sum_r partial[r][c] is
        # the ordered slab sum of `rows` parts of c floats; flag 2 keeps the serial order of cn_sum_rows_into for any row count
        st = _SINK
        cur0 = torch.cuda.current_stream()
        if all(cur0 != s_ for s_ in st.setdefault("touched", [])):
            st["touched"].append(cur0)
        st.setdefault("slabs", []).append((partial, dst, rows, c, 1 | 2))
        return
    fn = lambda: check(lib.cn_sum_rows_into(_fptr(partial), _fptr(dst), rows, c, int(accumulate), _stream()), "cn_sum_rows_into")
    if side:
        _sink_run(fn, (partial,))
    else:
        fn()


def bias_grad(gy, sink=None):
    """sum over every axis but the last of gy (the bias gradient of a linear / convolution layer without fused activation)."""
    gy = _c(gy)
    if sink is None:
        return nc_reduce(gy, None, want_dot=False, per_channel=True)[0].reshape(-1)
    c = gy.shape[-1]
    rows = gy.numel() // c
    if rows <= 64 and gy.dtype == torch.float32 and _SINK is not None and DEFER_SLAB_SUMS:
        # a dense layer's bias gradient over a batch of a few rows: the rows ARE the slabs of the pass' grouped reduction
        # (grad_sink.join) -- no launch of its own
        sum_rows_into(gy.reshape(rows, c), sink)
        return None
    rep = _partial_rows(rows)
    part = zero_pool_alloc((rep, c), gy.device)
    flags = 16
    if part is None:
        part, flags = torch.empty((rep, c), device=gy.device, dtype=torch.float32), 0
    check(lib.cn_nc_reduce(_ptr(gy), None, _ptr(part), None, rep, rows // rep, c, flags, 0.0, _dt(gy), _stream()), "cn_nc_reduce")
    sum_rows_into(part, sink)
    return None


def axpby(x, y, a, b):
    x, y = _unify(x, y)
    out = torch.empty_like(x)
    check(lib.cn_axpby(_ptr(x), _ptr(y), _ptr(out), x.numel(), a, b, _dt(x), _stream()), "cn_axpby")
    return out


def mul(x, y):
    x, y = _unify(x, y)
    out = torch.empty_like(x)
    check(lib.cn_mul(_ptr(x), _ptr(y), _ptr(out), x.numel(), _dt(x), _stream()), "cn_mul")
    return out


def sqdiff_sum(a, b, scale):
    a, b = _unify(a, b)
    out = torch.zeros((1,), device=a.device, dtype=torch.float32)
    check(lib.cn_sqdiff_sum(_ptr(a), _ptr(b), _ptr(out), a.numel(), scale, _dt(a), _stream()), "cn_sqdiff_sum")
    return out


def row_sumsq(x):
    x = f32(x)
    n = x.shape[0]
    out = torch.empty((n,), device=x.device, dtype=torch.float32)
    check(lib.cn_row_sumsq(_ptr(x), _ptr(out), n, x.numel() // n, _stream()), "cn_row_sumsq")
    return out


def row_scale(x, s, k):
    out = torch.empty_like(x)
    check(lib.cn_row_scale(_ptr(x), _fptr(s), _ptr(out), x.shape[0], x.numel() // x.shape[0], k, _dt(x), _stream()), "cn_row_scale")
    return out


def row_scale_diff(a, b, s, k):
    """(a - b) * s[row] * k in one pass (the backward of sqdiff_sum: s = the incoming scalar gradient, one row)."""
    a, b = _unify(a, b)
    out = torch.empty_like(a)
    check(lib.cn_row_scale_diff(_ptr(a), _ptr(b), _fptr(s), _ptr(out), s.numel(), a.numel() // s.numel(), k, _dt(a), _stream()), "cn_row_scale_diff")
    return out


def maxpool2_bwd_relu(x, gy, target=None, s=None, k=0.0):
    """Backward of ReLU -> MaxPooling2D(2, 2) in one pass (cn_maxpool2_bwd_act): (routed gy [+ (x - target) s k]) relu'(x) for the ReLU
    output x that was pooled; None where the kernel does not take the shape (the caller then runs the two passes)."""
    n, h, w, c = x.shape
    if target is None:
        x, gy = _unify(x, gy)
    else:
        x, gy, target = _unify(x, gy, target)
    if s is not None and s.numel() not in (1, n):
        return None
    gx = torch.empty_like(x)
    rc = lib.cn_maxpool2_bwd_act(_ptr(x), _ptr(_c(gy)), _ptr(target), _fptr(s), k, _ptr(gx), n, h, w, c, 1 if s is None else s.numel(), _dt(x),
                                 _stream())
    if rc == CN_EUNSUPPORTED:
        return None
    check(rc, "cn_maxpool2_bwd_act")
    if BRANCH_LOG is not None:                     # (test instrument: the pool's arg-max decisions and the ReLU mask, as the two passes log them)
        BRANCH_LOG.append(("pool", x.detach().float().cpu(), (2, 2, 0)))
    _log_mask(x, ACT_RELU)
    return gx


def tap_bwd(y, target, g, s, k, act, slope=0.0):
    """(g + (y - target) s[row] k) act'(y) in one pass (cn_tap_bwd); g may be None; s: one scale per row (s.numel() rows)."""
    if g is None:
        y, target = _unify(y, target)
    else:
        y, target, g = _unify(y, target, _c(g))
    _log_mask(y, act)
    out = torch.empty_like(y)
    check(lib.cn_tap_bwd(_ptr(y), _ptr(target), _ptr(g), _fptr(s), _ptr(out), s.numel(), y.numel() // s.numel(), k, act, slope, _dt(y),
                         _stream()), "cn_tap_bwd")
    return out


def masked_diff(a, b, mask):
    a, b = f32(a), f32(b)
    out = torch.empty_like(a)
    check(lib.cn_masked_diff(_ptr(a), _ptr(b), _ptr(mask), _ptr(out), mask.numel(), a.shape[-1], _stream()), "cn_masked_diff")
    return out


def maxpool_fwd(x, k, s, pad):
    n, h, w, c = x.shape
    oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    y = torch.empty((n, oh, ow, c), device=x.device, dtype=x.dtype)
    check(lib.cn_maxpool_fwd(_ptr(x), _ptr(y), n, h, w, c, k, s, pad, _dt(x), _stream()), "cn_maxpool_fwd")
    return y


def avgpool3_same(x):
    """AveragePooling2D((3, 3), strides 1, padding "same"); padding cells are not counted."""
    n, h, w, c = x.shape
    x = _c(x)
    y = torch.empty_like(x)
    check(lib.cn_avgpool3_same(_ptr(x), _ptr(y), n, h, w, c, _dt(x), _stream()), "cn_avgpool3_same")
    return y


def maxpool_bwd(x, gy, k, s, pad):
    n, h, w, c = x.shape
    x, gy = _unify(x, gy)
    if BRANCH_LOG is not None:                     # the kernel's decision: first maximum of a window in row-major order
        BRANCH_LOG.append(("pool", x.detach().float().cpu(), (k, s, pad)))
    gx = torch.empty_like(x)
    check(lib.cn_maxpool_bwd(_ptr(x), _ptr(gy), _ptr(gx), n, h, w, c, k, s, pad, _dt(x), _stream()), "cn_maxpool_bwd")
    return gx


def chan_affine3_fwd(x, perm, scale, off):
    x = f32(x)
    y = torch.empty_like(x)
    check(lib.cn_chan_affine3_fwd(_ptr(x), _ptr(y), x.numel() // 3, (ctypes.c_int * 3)(*perm), scale,
                                  (ctypes.c_float * 3)(*off), _stream()), "cn_chan_affine3_fwd")
    return y


def chan_affine3_bwd(gy, perm, scale):
    gy = f32(gy)
    gx = torch.empty_like(gy)
    check(lib.cn_chan_affine3_bwd(_ptr(gy), _ptr(gx), gy.numel() // 3, (ctypes.c_int * 3)(*perm), scale, _stream()),
          "cn_chan_affine3_bwd")
    return gx


def gan_loss_fwd(s, label):
    out = torch.empty((1,), device=s.device, dtype=torch.float32)
    check(lib.cn_gan_loss_fwd(_ptr(s), _ptr(out), s.numel(), label, _stream()), "cn_gan_loss_fwd")
    return out


def gan_loss_bwd(s, gout, label):
    gs = torch.empty_like(s)
    check(lib.cn_gan_loss_bwd(_ptr(s), _ptr(gout), _ptr(gs), s.numel(), label, _stream()), "cn_gan_loss_bwd")
    return gs


def gemm_rows_grouped(jobs):
    """cn_gemm_rows_grouped: every job (a, b, c, bias | None, mask | None, trans_b, act, slope, accumulate) -- c = epi(a op(b) + bias),
    a (m <= 32, k), contiguous fp32 2-D tensors -- in ONE launch."""
    arr = (CnRowsJob * len(jobs))()
    for q, (a, b, c, bias, mask, tb, act, slope, acc) in zip(arr, jobs):
        q.a, q.b, q.c = _fptr(a).value, _fptr(b).value, _fptr(c).value
        q.bias = None if bias is None else _fptr(bias).value
        q.mask = None if mask is None else _fptr(mask).value
        q.m, q.k = a.shape
        q.n = b.shape[0] if tb else b.shape[1]
        q.lda, q.ldb, q.ldc = a.shape[1], b.shape[1], c.shape[1]
        q.tb, q.act, q.accumulate, q.slope = int(tb), int(act), int(acc), float(slope)
    check(lib.cn_gemm_rows_grouped(arr, len(jobs), _stream()), "cn_gemm_rows_grouped")


def gan_loss_grouped(scores, labels, gouts=None, backward=False):
    """cn_gan_loss_grouped: the GAN losses of several heads (forward: one (H,) tensor of scalars) or their score gradients
    (backward: a list shaped like `scores`; gouts[j] = the cotangent scalar of head j or None) in ONE launch."""
    jobs = (CnGanJob * len(scores))()
    if backward:
        outs = [torch.empty_like(s_) for s_ in scores]
    else:
        res = torch.empty(len(scores), device=scores[0].device, dtype=torch.float32)
        outs = [res[j:j + 1] for j in range(len(scores))]
    for q, s_, o, lab, j in zip(jobs, scores, outs, labels, range(len(scores))):
        q.s, q.out, q.n, q.label = _fptr(s_).value, o.data_ptr(), s_.numel(), float(lab)
        q.gout = gouts[j].data_ptr() if (backward and gouts[j] is not None) else None
    check(lib.cn_gan_loss_grouped(jobs, len(scores), int(backward), _stream()), "cn_gan_loss_grouped")
    return outs if backward else res


def euler_matrix(angles):
    a = _c(f32(angles)).reshape(-1, 3)
    out = torch.empty((a.shape[0], 3, 3), device=a.device, dtype=torch.float32)
    check(lib.cn_euler_matrix(_ptr(a), _ptr(out), a.shape[0], _stream()), "cn_euler_matrix")
    return out


def euler_matrix_bwd(angles, grot):
    a, g = _c(f32(angles)).reshape(-1, 3), _c(f32(grot)).reshape(-1, 9)
    out = torch.empty_like(a)
    check(lib.cn_euler_matrix_bwd(_ptr(a), _ptr(g), _ptr(out), a.shape[0], _stream()), "cn_euler_matrix_bwd")
    return out


def rotate3d_fwd(grid, rot):
    keep = grid.dtype
    grid = f32(grid)                           # fp32 kernel; a bf16 grid is converted on the way in and out
    out = torch.empty_like(grid)
    n, g, c = grid.shape[0], grid.shape[1], grid.shape[-1]
    check(lib.cn_rotate3d_fwd(_ptr(grid), _fptr(rot), _ptr(out), n, g, c, _stream()), "cn_rotate3d_fwd")
    return cast(out, keep)


def rotate3d_bwd(grid, rot, gout, need_rot):
    n, g, c = grid.shape[0], grid.shape[1], grid.shape[-1]
    keep = grid.dtype
    grid, gout = f32(grid), f32(gout)
    ggrid = torch.empty_like(grid)
    grot = torch.empty((n, 3, 3), device=grid.device, dtype=torch.float32) if need_rot else None
    check(lib.cn_rotate3d_bwd(_ptr(grid), _fptr(rot), _ptr(gout), _ptr(ggrid), _ptr(grot), n, g, c, _stream()),
          "cn_rotate3d_bwd")
    return cast(ggrid, keep), grot


def adam_step(theta, grad, m, v, ema, lr_t, beta1, beta2, eps, ema_alpha=0.999):
    """lr_t: a 1-element float32 CUDA tensor (read by the kernel at execution time)."""
    check(lib.cn_adam_step(_ptr(theta), _ptr(grad), _ptr(m), _ptr(v), _ptr(ema), theta.numel(), _ptr(lr_t), beta1, beta2,
                           eps, ema_alpha, _stream()), "cn_adam_step")


def ema_step(ema, theta, alpha):
    check(lib.cn_ema_step(_ptr(ema), _ptr(theta), theta.numel(), alpha, _stream()), "cn_ema_step")


def gather_images_u8(pool, idx, flip):
    n = idx.numel()
    _, h, w, c = pool.shape
    out = torch.empty((n, h, w, c), device=pool.device, dtype=torch.float32)
    check(lib.cn_gather_images_u8(_ptr(pool), _ptr(idx), _ptr(flip), _ptr(out), n, h, w, c, _stream()),
          "cn_gather_images_u8")
    return out


def to_uint8(x):
    x = f32(x)
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    check(lib.cn_to_uint8(_ptr(x), _ptr(out), x.numel(), _stream()), "cn_to_uint8")
    return out


_PROF = {"on": False, "saved_flops": 0.0}


def prof_enable(on):
    _PROF["on"] = bool(on)
    check(lib.cn_prof_enable(int(on)), "cn_prof_enable")


def prof_reset():
    _PROF["saved_flops"] = 0.0
    check(lib.cn_prof_reset(), "cn_prof_reset")


def prof_note_saved(flops):
    """Multiply-adds a direct convolution would have issued and the algorithm in use does not (Winograd, parity-class
    collapse of upsample-folded layers): bench.py reports them next to the issued work, never inside `roofline.frac`."""
    if _PROF["on"]:
        _PROF["saved_flops"] += flops


def prof_saved_flops():
    return _PROF["saved_flops"]


def wino_saved_flops(g):
    """direct 3x3 count (border taps included: an upper bound) minus the 16 products per 2x2 tile issued"""
    tiles = g.n * ((g.in_h + 1) // 2) * ((g.in_w + 1) // 2)
    return 2.0 * g.cin * g.cout * (9.0 * g.n * g.in_h * g.in_w - 16.0 * tiles)


def upfold_saved_flops(g):
    """direct count of the conv on the x2-upsampled grid minus the parity-class form: (2/3)^nd of it for k3, (2.5/4)^nd for k4"""
    ks = [k for k in (g.k_d, g.k_h, g.k_w)][3 - g.nd:]
    ratio = 1.0
    for k in ks:
        ratio *= (2.0 / 3.0) if k == 3 else (2.5 / 4.0)
    direct = 2.0 * g.n * g.out_d * g.out_h * g.out_w * g.k_d * g.k_h * g.k_w * g.cin * g.cout
    return direct * (1.0 - ratio)


PROF_FAMILIES = ["igemm_fwd<128x128>", "igemm_fwd<128x64>", "igemm_fwd<64x64>", "igemm_fwd<128x32>", "igemm_fwd<128x96>",
                 "igemm_wgrad<128x128>", "igemm_wgrad<128x96>", "igemm_wgrad<64x64>", "igemm_wgrad<128x32>", "wino_fwd", "c3_fwd",
                 "s2_image_dgrad", "thin / up2k4_rgb", "c3_wgrad", "igemm_bf16", "igemm_bf16_wgrad", "igemm_wgrad<256x64>", "wgrad slab sums (grouped)"]


def prof_collect_by_family():
    """{family: {"launches", "ms", "gflop", "algorithmic_mb"}} of the launches recorded since prof_reset()."""
    n = (ctypes.c_int * 32)()
    ms, fl, by = (ctypes.c_double * 32)(), (ctypes.c_double * 32)(), (ctypes.c_double * 32)()
    check(lib.cn_prof_collect_by_family(n, ms, fl, by), "cn_prof_collect_by_family")
    out = {}
    for i in range(32):
        if n[i]:
            name = PROF_FAMILIES[i] if i < len(PROF_FAMILIES) else "other"
            out[name] = {"launches": int(n[i]), "ms": ms[i], "gflop": fl[i] / 1e9, "algorithmic_mb": by[i] / 1e6}
    return out


def prof_collect():
    n, ms, fl = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
    check(lib.cn_prof_collect(ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)), "cn_prof_collect")
    return n.value, ms.value, fl.value


if os.environ.get("CN_DETERMINISTIC") == "1":
    set_deterministic(True)
