"""Second, independent restatement of the forward ops in explicit NumPy (loops over
taps / voxels), float64.  Its only job is to cross-check oracle/ref_ops.py at tiny
shapes (SURVEY.md 8c "independent double implementation").  TEST INFRASTRUCTURE.
"""
import math
import numpy as np


def same_pad(n, k, s):
    out = int(math.ceil(n / s))
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2, out


def conv_same(x, w, b=None, stride=1):
    """x (N,*sp,Cin) ; w (*k,Cin,Cout).  Output position o reads input o*stride - pad_lo + tap."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    nd = x.ndim - 2
    sp, ks = x.shape[1:1 + nd], w.shape[:nd]
    pads = [same_pad(sp[i], ks[i], stride) for i in range(nd)]
    xp = np.pad(x, [(0, 0)] + [(p[0], p[1]) for p in pads] + [(0, 0)])
    osz = [p[2] for p in pads]
    y = np.zeros((x.shape[0], *osz, w.shape[-1]))
    for tap in np.ndindex(*ks):
        sl = tuple(slice(tap[i], tap[i] + (osz[i] - 1) * stride + 1, stride) for i in range(nd))
        patch = xp[(slice(None),) + sl]                       # (N,*osz,Cin)
        y += patch @ w[tap]
    if b is not None:
        y += b
    return y


def upsample2(x):
    for ax in range(1, x.ndim - 1):
        x = np.repeat(x, 2, axis=ax)
    return x


def leaky_relu(x, a):
    return np.where(x > 0, x, a * x)


def layer_norm_spatial(x, eps=1e-3):
    axes = tuple(range(1, x.ndim - 1))
    mu = x.mean(axis=axes, keepdims=True)
    var = x.var(axis=axes, keepdims=True)
    return (x - mu) / np.sqrt(var + eps)


def instance_norm(x, gamma, beta, eps=1e-3):
    axes = tuple(range(1, x.ndim - 1))
    mu = x.mean(axis=axes, keepdims=True)
    sd = x.std(axis=axes, keepdims=True) + eps
    return (x - mu) / sd * gamma + beta


def layer_style(x, eps=1e-6):
    axes = tuple(range(1, x.ndim - 1))
    mu = x.mean(axis=axes, keepdims=True)
    sd = np.sqrt(((x - mu) ** 2).mean(axis=axes, keepdims=True) + eps)
    return mu, sd


def euler_angles_to_matrix(a):
    a = np.asarray(a, np.float64).reshape(-1, 3)
    out = np.zeros((a.shape[0], 3, 3))
    for n, (ax, ay, az) in enumerate(a):
        s0, s1, s2 = math.sin(ax), math.sin(ay), math.sin(az)
        c0, c1, c2 = math.cos(ax), math.cos(ay), math.cos(az)
        out[n] = [[c2 * c1, -s2, c2 * s1],
                  [s0 * s1 + c0 * c1 * s2, c0 * c2, c0 * s2 * s1 - c1 * s0],
                  [c1 * s0 * s2 - c0 * s1, c2 * s0, c0 * c1 + s0 * s1 * s2]]
    return out


def transform_3d_grid(grid, transform):
    """Per-voxel loop version of transform_3d_grid_tf."""
    grid = np.asarray(grid, np.float64)
    n, g = grid.shape[0], grid.shape[1]
    ctr = (g - 1) / 2
    out = np.zeros_like(grid)
    for b in range(n):
        R = transform[b]
        for i in range(g):
            for j in range(g):
                for k in range(g):
                    q = R @ (np.array([i, j, k], np.float64) - ctr) + ctr
                    q = np.clip(q, 0, g - 1)
                    f = np.clip(np.floor(q), 0, g - 1)
                    c = np.clip(f + 1, 0, g - 1)
                    d = q - f
                    f, c = f.astype(int), c.astype(int)
                    c00 = grid[b, f[0], f[1], f[2]] * (1 - d[0]) + grid[b, c[0], f[1], f[2]] * d[0]
                    c01 = grid[b, f[0], f[1], c[2]] * (1 - d[0]) + grid[b, c[0], f[1], c[2]] * d[0]
                    c10 = grid[b, f[0], c[1], f[2]] * (1 - d[0]) + grid[b, c[0], c[1], f[2]] * d[0]
                    c11 = grid[b, f[0], c[1], c[2]] * (1 - d[0]) + grid[b, c[0], c[1], c[2]] * d[0]
                    c0 = c00 * (1 - d[1]) + c10 * d[1]
                    c1 = c01 * (1 - d[1]) + c11 * d[1]
                    out[b, i, j, k] = c0 * (1 - d[2]) + c1 * d[2]
    return out


def maxpool2(x):
    n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).max(axis=(2, 4))


def softplus(x):
    return np.logaddexp(0.0, x)
