"""Oracle ops: torch-CPU restatement of the TF/Keras ops the reference invokes.

TEST INFRASTRUCTURE (see oracle/__init__.py).  All tensors are channels-last
like the reference: (N, H, W, C) / (N, D, H, W, C).  Kernels use the Keras
layouts (kh, kw, cin, cout) / (kd, kh, kw, cin, cout) / Dense (in, out).
[TF-2.1] marks semantics that live in tensorflow 2.1 (not vendored).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# padding / conv   [TF-2.1] SAME padding rule R2
# ----------------------------------------------------------------------------
def same_pad(in_size, k, s):
    """total = max((ceil(in/s)-1)*s + k - in, 0); lo = total//2; hi = total-lo."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    lo = total // 2
    return lo, total - lo, out


def _to_cf(x):   # channels-last -> channels-first
    nd = x.dim() - 2
    return x.permute(0, nd + 1, *range(1, nd + 1))


def _to_cl(x):   # channels-first -> channels-last
    nd = x.dim() - 2
    return x.permute(0, *range(2, nd + 2), 1)


# ----------------------------------------------------------------------------
# bf16-storage emulation (tests of the bf16 compute path, BASELINE.json configs[2]): inside `with bf16_storage():` the network
# restatements of ref_nets.py round every tensor the product STORES in bf16 -- activations with more than 4 channels at the
# product's kernel boundaries (`stored`) and the filters of the convolutions that run on the bf16 matrix cores (`conv_same`: both
# channel counts multiples of 8) -- to bf16 and continue in float64.  What remains between product and oracle is then fp32
# accumulation, the rounding of pre-summed class filters and values that sit on a bf16 rounding boundary.
# ----------------------------------------------------------------------------
_BF16_STORAGE = False


class bf16_storage:
    def __enter__(self):
        global _BF16_STORAGE
        self.prev, _BF16_STORAGE = _BF16_STORAGE, True

    def __exit__(self, *exc):
        global _BF16_STORAGE
        _BF16_STORAGE = self.prev


def _bf16_round(t):
    return (t.detach().to(torch.float32).to(torch.bfloat16).to(t.dtype) - t.detach()) + t      # (straight-through for autograd)


def stored(x):
    """x as the product keeps it in HBM between two kernels (identity outside bf16_storage)."""
    return _bf16_round(x) if (_BF16_STORAGE and x.shape[-1] > 4) else x


def conv_same(x, w, b=None, stride=1):
    """keras.layers.Conv2D/Conv3D(padding="same") -- cross-correlation, asymmetric
    SAME padding (building_blocks.py:29,65,91; hologan_generator.py:50-56,101)."""
    if _BF16_STORAGE and w.shape[-2] % 8 == 0 and w.shape[-1] % 8 == 0:
        w = _bf16_round(w)
    nd = x.dim() - 2
    ks = w.shape[:nd]
    spatial = x.shape[1:1 + nd]
    pads = []
    for i in reversed(range(nd)):            # F.pad wants last dim first
        lo, hi, _ = same_pad(spatial[i], ks[i], stride)
        pads += [lo, hi]
    xc = F.pad(_to_cf(x), pads)
    wc = w.permute(nd + 1, nd, *range(nd))   # (cout, cin, *k)
    fn = F.conv2d if nd == 2 else F.conv3d
    y = fn(xc, wc, b, stride=stride)
    return _to_cl(y)


def conv_valid_padded(x, w, b, stride, pad):
    """ZeroPadding2D(pad) + Conv2D(padding="valid") (keras ResNet50 conv1)."""
    xc = F.pad(_to_cf(x), [pad, pad, pad, pad])
    wc = w.permute(3, 2, 0, 1)
    return _to_cl(F.conv2d(xc, wc, b, stride=stride))


def upsample2(x):
    """keras UpSampling2D()/UpSampling3D(): size 2, nearest (R6)."""
    nd = x.dim() - 2
    for ax in range(1, 1 + nd):
        x = x.repeat_interleave(2, dim=ax)
    return x


class BranchControl:
    """Test instrument, not part of the restated algorithm.  A piecewise-linear activation whose input lies within the
    product's fp32 rounding of zero may take the other branch on the GPU than in this float64 oracle; the value is continuous
    there, but the derivative mask differs, which moves every upstream gradient by ~1/sqrt(#elements).  Three modes:
    * `record`: the activations list their near-zero (non-zero) inputs per call -- flat index, |x| / mean|x|, and, after a
      backward pass, the magnitude of the gradient that reached that element;
    * `flips` = {(call number, flat index)}: the listed elements take the other branch (tests/test_nets_gpu.py: check_grads
      attributes a deviation to named decisions with these two);
    * `forced` = the product's own decisions (confignet_amd.ops.branch_log: the boolean mask `t > 0` of every tensor its
      backward pass took a LeakyReLU / ReLU derivative from, and the fp32 input of every max-pool it differentiated): an
      activation (max-pool) of this oracle that a gradient can reach looks up the logged mask of its size that agrees with its
      own decisions everywhere except on near-zero inputs (near-ties), and TAKES it.  With every decision forced, product and
      oracle gradients differ by summation error only -- the whole-step chains are then held at the single networks' 5e-3.
      `forced_report()` lists what was forced and what found no match."""
    active = False
    record = False
    delta = 1e-4
    calls = 0
    cand = {}            # call number -> (flat indices, margins, gradient magnitudes)
    flips = frozenset()
    forced = None        # {"act": {numel: [{"mask", "pop"}]}, "pool": {key: [fp32 input]}}
    force_margin = 2e-3  # a forced decision may differ from the oracle's own only where |x| < force_margin * mean|x|
    report = None

    @classmethod
    def start(cls, record=False, flips=(), forced=None):
        cls.active, cls.record, cls.flips, cls.calls, cls.cand = True, record, frozenset(flips), 0, {}
        cls.forced, cls.report = None, {"forced_calls": 0, "forced_decisions": 0, "max_margin": 0.0, "unmatched": []}
        if forced is not None:
            acts, pools = {}, {}
            for entry in forced:
                if entry[0] == "act":
                    mk = entry[1].reshape(-1)
                    acts.setdefault(mk.numel(), []).append({"mask": mk, "pop": int(mk.sum())})
                else:
                    x32, key = entry[1], entry[2]
                    pools.setdefault((tuple(x32.shape), tuple(key)), []).append(x32)
            cls.forced = {"act": acts, "pool": pools}

    @classmethod
    def stop(cls):
        cls.active = cls.record = False
        cls.flips = frozenset()
        cls.forced = None

    @classmethod
    def candidates(cls):
        """[(call number, flat index, margin, gradient magnitude)] of the last recorded run."""
        out = []
        for cid, (idx, mar, infl) in cls.cand.items():
            out += [(cid, int(i), float(m), float(g)) for i, m, g in zip(idx, mar, infl)]
        return out

    @classmethod
    def forced_report(cls):
        """What the last forced run did; `unmatched` holds only calls that a gradient actually reached."""
        r = dict(cls.report)
        r["unmatched"] = [u for u in r["unmatched"] if u["reached"][0]]
        return r

    @classmethod
    def _unmatched(cls, y, what, **info):
        reached = [False]
        cls.report["unmatched"].append(dict(info, what=what, reached=reached))
        if y.requires_grad:
            def note(g, reached=reached):
                reached[0] = reached[0] or bool((g != 0).any())
            y.register_hook(note)

    @classmethod
    def forced_mask(cls, xd):
        """The logged product mask for this activation input, or (None, info)."""
        flat = xd.reshape(-1)
        own = flat > 0
        pop = int(own.sum())
        best = None
        for e in sorted(cls.forced["act"].get(flat.numel(), ()), key=lambda e: abs(e["pop"] - pop)):
            if best is not None and abs(e["pop"] - pop) >= best[0]:
                break
            d = int((e["mask"] != own).sum())
            if best is None or d < best[0]:
                best = (d, e)
            if d == 0:
                break
        if best is None or best[0] > 0.01 * flat.numel():
            # the product may have run this activation on a STACKED batch (several of this oracle's calls in one launch: the
            # samples of a stacked tensor are contiguous blocks of its mask): look for the block among the larger logged masks
            for size, entries in cls.forced["act"].items():
                if size <= flat.numel() or size % flat.numel():
                    continue
                for e in entries:
                    blocks = e["mask"].reshape(size // flat.numel(), flat.numel())
                    dist = (blocks != own.unsqueeze(0)).sum(dim=1)
                    k = int(dist.argmin())
                    if best is None or int(dist[k]) < best[0]:
                        best = (int(dist[k]), {"mask": blocks[k], "pop": int(blocks[k].sum())})
        if best is None or best[0] > 0.01 * flat.numel():
            # ... or the other way round: this oracle call works on a stack whose halves the product ran as separate launches
            # (the second-stage generator step runs the latent regressor per branch): assemble the mask from the logged
            # masks of the contiguous sample blocks
            for k in (2, 4):
                if flat.numel() % k or not cls.forced["act"].get(flat.numel() // k):
                    continue
                parts, total = [], 0
                for blk in own.reshape(k, -1):
                    bp, bb = int(blk.sum()), None
                    for e in sorted(cls.forced["act"][blk.numel()], key=lambda e: abs(e["pop"] - bp)):
                        if bb is not None and abs(e["pop"] - bp) >= bb[0]:
                            break
                        d = int((e["mask"] != blk).sum())
                        if bb is None or d < bb[0]:
                            bb = (d, e["mask"])
                        if d == 0:
                            break
                    parts.append(bb[1])
                    total += bb[0]
                if best is None or total < best[0]:
                    m = torch.cat(parts)
                    best = (total, {"mask": m, "pop": int(m.sum())})
        if best is None:
            return None, {"numel": flat.numel(), "why": "no logged mask of this size"}
        d, e = best
        if d:
            idx = torch.nonzero(e["mask"] != own).reshape(-1)
            margin = float(flat[idx].abs().max() / (flat.abs().mean() + 1e-300))
            if margin > cls.force_margin:
                return None, {"numel": flat.numel(), "why": "nearest logged mask differs in %d decisions, margin %.2e" % (d, margin)}
            cls.report["max_margin"] = max(cls.report["max_margin"], margin)
        cls.report["forced_calls"] += 1
        cls.report["forced_decisions"] += d
        return e["mask"], None


def _branch_act(x, neg):
    """max-like activation as x * (constant derivative mask): 1 where x > 0, `neg` elsewhere -- the same values and the same
    first and second derivatives as where(x > 0, x, neg * x); the mask form lets BranchControl name single branch decisions."""
    xd = x.detach()
    m = torch.where(xd > 0, torch.ones((), dtype=x.dtype), torch.full((), neg, dtype=x.dtype))
    bc = BranchControl
    if not bc.active:
        return x * m
    cid, bc.calls = bc.calls, bc.calls + 1
    if bc.forced is not None:
        if not x.requires_grad:
            return x * m
        pm, info = bc.forced_mask(xd)
        if pm is None:
            y = x * m
            bc._unmatched(y, "activation", call=cid, shape=tuple(x.shape), **info)
            return y
        m = torch.where(pm.reshape(x.shape), torch.ones((), dtype=x.dtype), torch.full((), neg, dtype=x.dtype))
        return x * m
    for c, i in bc.flips:
        if c == cid:
            mf = m.reshape(-1)
            mf[i] = neg if float(mf[i]) == 1.0 else 1.0
    y = x * m
    if bc.record:
        ax = xd.abs().reshape(-1)
        scale = float(ax.mean()) + 1e-300
        idx = torch.nonzero((ax < bc.delta * scale) & (ax > 0)).reshape(-1)
        if idx.numel():
            infl = torch.zeros(idx.numel(), dtype=torch.float64)
            bc.cand[cid] = (idx, ax[idx] / scale, infl)
            if y.requires_grad:
                def note(g, idx=idx, infl=infl):
                    infl.add_(g.detach().reshape(-1)[idx].abs().double())
                y.register_hook(note)
    return y


def leaky_relu(x, alpha):
    """keras.layers.LeakyReLU() default alpha=0.3; tf.nn.leaky_relu default 0.2 (R1)."""
    return _branch_act(x, alpha)


def relu(x):
    return _branch_act(x, 0.0)



def dense(x, w, b=None):
    y = x @ w
    return y if b is None else y + b


def mlp_simple(x, weights, alpha, alpha_last=None):
    """MLPSimple (building_blocks.py:152-173): (Dense -> nonlin) x (L-1), Dense."""
    n_layers = len(weights) // 2
    for i in range(n_layers):
        x = dense(x, weights[2 * i], weights[2 * i + 1])
        if i < n_layers - 1:
            x = leaky_relu(x, alpha)
        elif alpha_last is not None:
            x = leaky_relu(x, alpha_last)
    return x


# ----------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------
def adain(x, z, mlp_weights, mlp_alpha=0.2):
    """AdaIn (building_blocks.py:114-149): LayerNormalization over the spatial axes,
    center=False, scale=False, eps 1e-3 [TF-2.1], biased variance, (x-mu)*rsqrt(var+eps);
    then x*(s+1)+b with [s, b] = MLP(z) split as (N, 2, C)."""
    nd = x.dim() - 2
    axes = tuple(range(1, 1 + nd))
    c = x.shape[-1]
    sb = mlp_simple(z, mlp_weights, mlp_alpha)
    sb = sb.reshape(-1, 2, *([1] * nd), c)
    mu = x.mean(dim=axes, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=axes, keepdim=True)
    xn = (x - mu) * torch.rsqrt(var + 1e-3)
    return xn * (sb[:, 0] + 1) + sb[:, 1]


def instance_norm(x, gamma, beta, eps=1e-3):
    """InstanceNormalization(axis=-1) (instance_normalization.py:108-131): eps is
    added to the *std*: (x-mean)/(std+eps)*gamma+beta, biased std."""
    axes = tuple(range(1, x.dim() - 1))
    mu = x.mean(dim=axes, keepdim=True)
    std = torch.sqrt(((x - mu) ** 2).mean(dim=axes, keepdim=True)) + eps
    return (x - mu) / std * gamma + beta


def layer_style(x, eps=1e-6):
    """get_layer_style (confignet_utils.py:147-159)."""
    axes = tuple(range(1, x.dim() - 1))
    mu = x.mean(dim=axes, keepdim=True)
    std = torch.sqrt(((x - mu) ** 2).mean(dim=axes, keepdim=True) + eps)
    return mu, std


# ----------------------------------------------------------------------------
# 3-D rotation
# ----------------------------------------------------------------------------
def euler_angles_to_matrix(angles):
    """confignet_utils.py:122-145."""
    a = angles.reshape(-1, 3)
    s, c = torch.sin(a), torch.cos(a)
    a11 = c[:, 2] * c[:, 1]
    a12 = -s[:, 2]
    a13 = c[:, 2] * s[:, 1]
    a21 = s[:, 0] * s[:, 1] + c[:, 0] * c[:, 1] * s[:, 2]
    a22 = c[:, 0] * c[:, 2]
    a23 = c[:, 0] * s[:, 2] * s[:, 1] - c[:, 1] * s[:, 0]
    a31 = c[:, 1] * s[:, 0] * s[:, 2] - c[:, 0] * s[:, 1]
    a32 = c[:, 2] * s[:, 0]
    a33 = c[:, 0] * c[:, 1] + s[:, 0] * s[:, 1] * s[:, 2]
    return torch.stack([a11, a12, a13, a21, a22, a23, a31, a32, a33], dim=-1).reshape(-1, 3, 3)


def transform_3d_grid(grid, transform):
    """transform_3d_grid_tf (confignet_utils.py:63-120): rigid rotation about the grid
    centre, clamp-to-edge, trilinear (x, then y, then z).  tf.floor has zero gradient;
    tf.clip_by_value passes gradient strictly inside the interval (R12)."""
    n, g = grid.shape[0], grid.shape[1]
    assert grid.shape[1] == grid.shape[2] == grid.shape[3]
    centre = (g - 1) / 2
    ar = torch.arange(g, dtype=grid.dtype)
    xs, ys, zs = torch.meshgrid(ar, ar, ar, indexing="ij")
    coords = torch.stack([xs.reshape(-1), ys.reshape(-1), zs.reshape(-1)])      # (3, P)
    tc = transform.to(grid.dtype) @ (coords - centre) + centre                  # (N, 3, P)
    # clip_by_value: gradient 1 strictly inside, 0 outside/at the clamp
    tc = torch.clamp(tc, 0, g - 1)
    fl = torch.clamp(torch.floor(tc.detach()), 0, g - 1)
    ce = torch.clamp(fl + 1, 0, g - 1)
    fi, ci = fl.long(), ce.long()
    bidx = torch.arange(n).reshape(n, 1).expand(n, g ** 3)

    def gat(ix, iy, iz):
        return grid[bidx, ix, iy, iz]                                            # (N, P, C)

    x0, y0, z0 = fi[:, 0], fi[:, 1], fi[:, 2]
    x1, y1, z1 = ci[:, 0], ci[:, 1], ci[:, 2]
    c000, c100 = gat(x0, y0, z0), gat(x1, y0, z0)
    c101, c001 = gat(x1, y0, z1), gat(x0, y0, z1)
    c010, c110 = gat(x0, y1, z0), gat(x1, y1, z0)
    c111, c011 = gat(x1, y1, z1), gat(x0, y1, z1)
    d = (tc - fl).unsqueeze(-1)                                                  # (N, 3, P, 1)
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    c00 = c000 * (1 - dx) + c100 * dx
    c01 = c001 * (1 - dx) + c101 * dx
    c10 = c010 * (1 - dx) + c110 * dx
    c11 = c011 * (1 - dx) + c111 * dx
    c0 = c00 * (1 - dy) + c10 * dy
    c1 = c01 * (1 - dy) + c11 * dy
    out = c0 * (1 - dz) + c1 * dz
    return out.reshape(grid.shape)


# ----------------------------------------------------------------------------
# pooling / misc (keras.applications VGG / ResNet50)
# ----------------------------------------------------------------------------
def maxpool(x, k, s, pad=0):
    """MaxPooling2D(k, strides=s) after an optional ZeroPadding2D(pad)."""
    xc = _to_cf(x)
    if pad:
        xc = F.pad(xc, [pad, pad, pad, pad])     # zero padding, as ZeroPadding2D does
    bc = BranchControl
    if bc.active and bc.forced is not None and x.requires_grad:
        return _to_cl(_forced_maxpool(x, xc, k, s, pad))
    return _to_cl(F.max_pool2d(xc, k, s))


def _forced_maxpool(x, xc, k, s, pad):
    """BranchControl(forced=...): the window winners of the PRODUCT (first maximum in row-major window order of its fp32 input,
    tests/test_ops_gpu.py pins that rule on tied inputs) instead of this float64 pass's own; a winner may differ only between
    near-tied window elements."""
    bc = BranchControl
    own, own_idx = F.max_pool2d(xc.detach(), k, s, return_indices=True)
    best = None
    logged = list(bc.forced["pool"].get((tuple(x.shape), (k, s, pad)), ()))
    if not logged:                               # (a stacked batch on the product's side: its samples one by one, see forced_mask)
        for (shp, key), xs in bc.forced["pool"].items():
            if key == (k, s, pad) and tuple(shp[1:]) == tuple(x.shape[1:]) and shp[0] > x.shape[0] and shp[0] % x.shape[0] == 0:
                for x32 in xs:
                    logged += list(x32.split(x.shape[0], dim=0))
    for x32 in logged:
        p32 = _to_cf(x32)
        if pad:
            p32 = F.pad(p32, [pad, pad, pad, pad])
        _, idx = F.max_pool2d(p32, k, s, return_indices=True)
        d = int((idx != own_idx).sum())
        if best is None or d < best[0]:
            best = (d, idx)
        if d == 0:
            break
    flat = xc.flatten(2)
    if best is None:
        y = F.max_pool2d(xc, k, s)
        bc._unmatched(y, "max-pool", shape=tuple(x.shape), why="no logged pool input of this shape")
        return y
    d, idx = best
    y = flat.gather(2, idx.flatten(2)).reshape(own.shape)
    if d:
        gap = float((own - y.detach()).abs().max() / (xc.detach().abs().mean() + 1e-300))
        if gap > bc.force_margin:
            y = F.max_pool2d(xc, k, s)
            bc._unmatched(y, "max-pool", shape=tuple(x.shape), why="nearest logged pool differs in %d winners, gap %.2e" % (d, gap))
            return y
        bc.report["max_margin"] = max(bc.report["max_margin"], gap)
    bc.report["forced_calls"] += 1
    bc.report["forced_decisions"] += d
    return y


def bn_inference(x, gamma, beta, mean, var, eps):
    """BatchNormalization in inference mode (R9)."""
    return (x - mean) * (gamma / torch.sqrt(var + eps)) + beta


def caffe_preprocess(x_m11):
    """(x+1)*127.5 -> keras preprocess_input mode "caffe" (R8): RGB->BGR flip of the
    channel axis, subtract (103.939, 116.779, 123.68)  (perceptual_loss.py:52-61,
    real_encoder.py:24-25)."""
    x = (x_m11 + 1) * 127.5
    x = x.flip(-1)
    return x - torch.tensor([103.939, 116.779, 123.68], dtype=x.dtype)


def vggface_preprocess(x_m11):
    """perceptual_loss.py:53-56: subtract (93.5940, 104.7624, 129.1863), no flip."""
    x = (x_m11 + 1) * 127.5
    return x - torch.tensor([93.5940, 104.7624, 129.1863], dtype=x.dtype)


# ----------------------------------------------------------------------------
# losses (losses.py)
# ----------------------------------------------------------------------------
def gan_g_loss(scores):
    """losses.py:7-8."""
    return F.softplus(-scores).mean()


def gan_d_loss(labels, scores):
    """losses.py:10-11."""
    return (labels * F.softplus(-scores) + (1.0 - labels) * F.softplus(scores)).mean()


def eye_loss(gt, gen, masks):
    """losses.py:13-18; masks uint8 (N,H,W)."""
    m = masks.to(gt.dtype)
    diff = (gt - gen) * m.unsqueeze(-1)
    per = (diff ** 2).sum(dim=(1, 2, 3)) / (1 + m.sum(dim=(1, 2)))
    return per.mean()


def r1_penalty(out, x):
    """gradient_regularization (losses.py:75-82): 10*0.5*mean_n sum (d sum(out)/dx)^2."""
    (g,) = torch.autograd.grad(out.sum(), x, create_graph=True)
    return 10 * 0.5 * (g ** 2).reshape(g.shape[0], -1).sum(dim=1).mean()


# ----------------------------------------------------------------------------
# optimizer [TF-2.1] Keras Adam (R10)
# ----------------------------------------------------------------------------
class KerasAdam:
    """theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps), eps=1e-7; ONE step counter
    per optimizer object shared by every variable it updates."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False):
        assert not amsgrad
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.t = 0
        self.state = {}

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        with torch.no_grad():
            for g, p in grads_and_vars:
                if g is None:
                    continue
                st = self.state.setdefault(id(p), [torch.zeros_like(p), torch.zeros_like(p)])
                st[0].mul_(self.b1).add_(g, alpha=1 - self.b1)
                st[1].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                p.sub_(lr_t * st[0] / (torch.sqrt(st[1]) + self.eps))


def glorot_uniform(rng, shape):
    """Keras default kernel initializer (R4): U(-l, l), l = sqrt(6/(fan_in+fan_out));
    fan_in = prod(k)*cin, fan_out = prod(k)*cout."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)
