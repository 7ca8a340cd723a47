"""Oracle of the training-time metrics (reference: confignet/metrics/inception_distance.py, metrics.py:201-214).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference computes FID / KID on features of
keras.applications.inception_v3.InceptionV3(include_top=False, weights="imagenet", pooling="avg") -- a third-party model that
is not vendored (tensorflow 2.1 / keras_applications 1.0.8, inception_v3.py).  Its published architecture is restated here
block by block in float64 torch (Conv2D(use_bias=False) -> BatchNormalization(scale=False, epsilon=1e-3, inference) -> ReLU,
TF SAME / VALID padding, MaxPooling2D((3,3), 2), AveragePooling2D((3,3), 1, "same") without counting padding cells), and
the host-side formulas with the libraries the reference calls (scipy.linalg.sqrtm, sklearn polynomial_kernel).
Parity unpinned against executed TF: `scripts/tf_pin_dump.py` also dumps the InceptionV3 weight shapes in get_weights() order;
the layer order is pinned on the published `model.summary()` of the first inception block (tests/test_metrics_cpu.py)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import ref_ops as O


class _Tape:
    """Records layers in creation order while the architecture function below runs (what the Keras functional API does)."""

    def __init__(self):
        self.layers = []          # dict(name, kind, inputs, **params)

    def new(self, kind, inputs, **kw):
        n = "L%d" % len(self.layers)
        self.layers.append(dict(name=n, kind=kind, inputs=list(inputs), **kw))
        return n


def _cbr(t, x, f, rows, cols, padding="same", strides=1):
    c = t.new("conv", [x], filters=f, k=(rows, cols), padding=padding, strides=strides)
    return t.new("act", [t.new("bn", [c])])


def inception_v3_layers():
    t = _Tape()
    x = t.new("input", [])
    x = _cbr(t, x, 32, 3, 3, "valid", 2)
    x = _cbr(t, x, 32, 3, 3, "valid")
    x = _cbr(t, x, 64, 3, 3)
    x = t.new("maxpool", [x])
    x = _cbr(t, x, 80, 1, 1, "valid")
    x = _cbr(t, x, 192, 3, 3, "valid")
    x = t.new("maxpool", [x])
    for pf in (32, 64, 64):
        a = _cbr(t, x, 64, 1, 1)
        b = _cbr(t, _cbr(t, x, 48, 1, 1), 64, 5, 5)
        c = _cbr(t, _cbr(t, _cbr(t, x, 64, 1, 1), 96, 3, 3), 96, 3, 3)
        d = _cbr(t, t.new("avgpool", [x]), pf, 1, 1)
        x = t.new("concat", [a, b, c, d])
    a = _cbr(t, x, 384, 3, 3, "valid", 2)
    c = _cbr(t, _cbr(t, _cbr(t, x, 64, 1, 1), 96, 3, 3), 96, 3, 3, "valid", 2)
    x = t.new("concat", [a, c, t.new("maxpool", [x])])
    for f in (128, 160, 160, 192):
        a = _cbr(t, x, 192, 1, 1)
        b = _cbr(t, _cbr(t, _cbr(t, x, f, 1, 1), f, 1, 7), 192, 7, 1)
        c = _cbr(t, x, f, 1, 1)
        for k, (r, cc) in zip((f, f, f, 192), ((7, 1), (1, 7), (7, 1), (1, 7))):
            c = _cbr(t, c, k, r, cc)
        d = _cbr(t, t.new("avgpool", [x]), 192, 1, 1)
        x = t.new("concat", [a, b, c, d])
    a = _cbr(t, _cbr(t, x, 192, 1, 1), 320, 3, 3, "valid", 2)
    b = _cbr(t, _cbr(t, _cbr(t, _cbr(t, x, 192, 1, 1), 192, 1, 7), 192, 7, 1), 192, 3, 3, "valid", 2)
    x = t.new("concat", [a, b, t.new("maxpool", [x])])
    for _ in range(2):
        a = _cbr(t, x, 320, 1, 1)
        b = _cbr(t, x, 384, 1, 1)
        b = t.new("concat", [_cbr(t, b, 384, 1, 3), _cbr(t, b, 384, 3, 1)])
        c = _cbr(t, _cbr(t, x, 448, 1, 1), 384, 3, 3)
        c = t.new("concat", [_cbr(t, c, 384, 1, 3), _cbr(t, c, 384, 3, 1)])
        d = _cbr(t, t.new("avgpool", [x]), 192, 1, 1)
        x = t.new("concat", [a, b, c, d])
    t.new("gap", [x])
    return t.layers


def model_layers_order(layers):
    """[TF-2.1] tensorflow/python/keras/engine/network.py _map_graph_network: depth of a layer = longest path to the output;
    model.layers = layers by decreasing depth, ties by the index given on first visit of a recursive walk from the output
    through each node's inbound tensors in order."""
    import sys
    by = {l["name"]: l for l in layers}
    seen = {}

    def visit(n):
        if n in seen:
            return
        seen[n] = len(seen)
        for i in by[n]["inputs"]:
            visit(i)
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(10000)
    try:
        visit(layers[-1]["name"])
    finally:
        sys.setrecursionlimit(old)
    depth = {layers[-1]["name"]: 0}
    for l in reversed(layers):                               # consumers are created after producers
        d = depth.setdefault(l["name"], 0)
        for i in l["inputs"]:
            depth[i] = max(depth.get(i, 0), d + 1)
    return sorted(layers, key=lambda l: (-depth[l["name"]], seen[l["name"]]))


def inception_weight_shapes():
    """Shapes of InceptionV3(include_top=False).get_weights(): per layer in model.layers order, Conv2D [kernel],
    BatchNormalization(scale=False) [beta, moving_mean, moving_variance]."""
    layers = inception_v3_layers()
    ch = {}
    for l in layers:
        if l["kind"] == "input":
            ch[l["name"]] = 3
        elif l["kind"] == "conv":
            l["cin"] = ch[l["inputs"][0]]
            ch[l["name"]] = l["filters"]
        elif l["kind"] == "concat":
            ch[l["name"]] = sum(ch[i] for i in l["inputs"])
        else:
            ch[l["name"]] = ch[l["inputs"][0]]
    shapes = []
    for l in model_layers_order(layers):
        if l["kind"] == "conv":
            shapes.append((l["k"][0], l["k"][1], l["cin"], l["filters"]))
        elif l["kind"] == "bn":
            shapes += [(ch[l["name"]],)] * 3
    return shapes


def _conv(x, w, padding, stride):
    if padding == "same":
        return O.conv_same(x, w, None, stride)
    return O._to_cl(F.conv2d(O._to_cf(x), w.permute(3, 2, 0, 1), None, stride=stride))


def _avgpool3_same(x):
    xc = O._to_cf(x)
    return O._to_cl(F.avg_pool2d(xc, 3, 1, padding=1, count_include_pad=False))


def inception_features(weights, x):
    """x: (N, H, W, 3) in [-1, 1] (after preprocess_input) -> (N, 2048); weights in get_weights() order."""
    layers = inception_v3_layers()
    w = {}
    it = iter(weights)
    for l in model_layers_order(layers):
        if l["kind"] == "conv":
            w[l["name"]] = [next(it)]
        elif l["kind"] == "bn":
            w[l["name"]] = [next(it), next(it), next(it)]
    v = {}
    for l in layers:
        k, ins = l["kind"], l["inputs"]
        if k == "input":
            v[l["name"]] = x
        elif k == "conv":
            v[l["name"]] = _conv(v[ins[0]], w[l["name"]][0], l["padding"], l["strides"])
        elif k == "bn":
            beta, mean, var = w[l["name"]]
            v[l["name"]] = (v[ins[0]] - mean) / torch.sqrt(var + 1e-3) + beta
        elif k == "act":
            v[l["name"]] = torch.relu(v[ins[0]])
        elif k == "maxpool":
            v[l["name"]] = O.maxpool(v[ins[0]], 3, 2)
        elif k == "avgpool":
            v[l["name"]] = _avgpool3_same(v[ins[0]])
        elif k == "concat":
            v[l["name"]] = torch.cat([v[i] for i in ins], dim=-1)
        elif k == "gap":
            v[l["name"]] = v[ins[0]].mean(dim=(1, 2))
    return v[layers[-1]["name"]]


def compute_FID(features_g, features_r):
    """inception_distance.py:30-45"""
    import scipy.linalg
    mean_g, mean_r = np.mean(features_g, axis=0), np.mean(features_r, axis=0)
    cov_g, cov_r = np.cov(features_g, rowvar=False), np.cov(features_r, rowvar=False)
    return np.linalg.norm(mean_g - mean_r) ** 2 + np.real(np.trace(cov_g + cov_r - 2 * scipy.linalg.sqrtm(np.dot(cov_g, cov_r))))


def compute_KID(features_g, features_r):
    """inception_distance.py:47-59"""
    from sklearn.metrics.pairwise import polynomial_kernel
    gg = polynomial_kernel(features_g, degree=3, coef0=1.0)
    rr = polynomial_kernel(features_r, degree=3, coef0=1.0)
    gr = polynomial_kernel(features_g, features_r, degree=3, coef0=1.0)
    m, n = features_g.shape[0], features_r.shape[0]
    return ((np.sum(gg) - np.sum(np.diagonal(gg))) / (m * (m - 1)) + (np.sum(rr) - np.sum(np.diagonal(rr))) / (n * (n - 1))
            - 2 * np.sum(gr) / (m * n))
