"""FID / KID (reference: confignet/metrics/inception_distance.py).

`InceptionFeatureExtractor` is keras.applications.inception_v3.InceptionV3(include_top=False, pooling="avg") [TF-2.1, not
vendored by the reference; restated from the published architecture, Szegedy et al. 2015 as implemented in
keras_applications/inception_v3.py] on the HIP kernels: every Conv2D(use_bias=False) -> BatchNormalization(scale=False,
eps 1e-3) -> ReLU triple is ONE convolution launch (the inference-mode normalisation folded into the filter and a bias,
ReLU in the epilogue), the pool branches are `cn_maxpool_fwd` / `cn_avgpool3_same`, the final pooling one `cn_nc_reduce`.
Weights are held in the Keras `get_weights()` order (per layer in `model.layers` order: kernel | beta, moving_mean,
moving_variance), so a `np.savez` of `InceptionV3(weights="imagenet").get_weights()` made on a TensorFlow machine loads with
`set_weights` / `load_keras_weights`.  The imagenet weights cannot be downloaded here: without a weights file the network
is He-initialised (seeded) and says so -- FID/KID values are then only comparable between runs of this code.

compute_FID / compute_KID are the reference's host-side formulas (the feature matrices are (n, 2048))."""
import numpy as np
import torch

from .. import ops
from ..nn import Net
from ..ops import ACT_RELU, ConvSpec


# ---- architecture as a layer graph (creation order of keras_applications/inception_v3.py) --------------------------------
class _Graph:
    def __init__(self):
        self.layers = []                       # (name, kind, inputs, params) in creation order

    def add(self, kind, inputs, **params):
        name = "%s_%d" % (kind, len(self.layers))
        self.layers.append((name, kind, list(inputs), params))
        return name

    def conv_bn(self, x, filters, rows, cols, padding="same", stride=1):
        c = self.add("conv", [x], filters=filters, kernel=(rows, cols), padding=padding, stride=stride)
        b = self.add("bn", [c])
        return self.add("relu", [b])


def inception_v3_graph():
    g = _Graph()
    x = g.add("input", [])
    x = g.conv_bn(x, 32, 3, 3, "valid", 2)
    x = g.conv_bn(x, 32, 3, 3, "valid")
    x = g.conv_bn(x, 64, 3, 3)
    x = g.add("maxpool", [x])
    x = g.conv_bn(x, 80, 1, 1, "valid")
    x = g.conv_bn(x, 192, 3, 3, "valid")
    x = g.add("maxpool", [x])
    for pool_filters in (32, 64, 64):                                   # mixed 0, 1, 2: 35 x 35
        b1 = g.conv_bn(x, 64, 1, 1)
        b5 = g.conv_bn(g.conv_bn(x, 48, 1, 1), 64, 5, 5)
        b3 = g.conv_bn(g.conv_bn(g.conv_bn(x, 64, 1, 1), 96, 3, 3), 96, 3, 3)
        bp = g.conv_bn(g.add("avgpool", [x]), pool_filters, 1, 1)
        x = g.add("concat", [b1, b5, b3, bp])
    b3 = g.conv_bn(x, 384, 3, 3, "valid", 2)                            # mixed 3: 17 x 17
    bd = g.conv_bn(g.conv_bn(g.conv_bn(x, 64, 1, 1), 96, 3, 3), 96, 3, 3, "valid", 2)
    x = g.add("concat", [b3, bd, g.add("maxpool", [x])])
    for f in (128, 160, 160, 192):                                      # mixed 4 .. 7
        b1 = g.conv_bn(x, 192, 1, 1)
        b7 = g.conv_bn(g.conv_bn(g.conv_bn(x, f, 1, 1), f, 1, 7), 192, 7, 1)
        bd = g.conv_bn(x, f, 1, 1)
        bd = g.conv_bn(g.conv_bn(g.conv_bn(g.conv_bn(bd, f, 7, 1), f, 1, 7), f, 7, 1), 192, 1, 7)
        bp = g.conv_bn(g.add("avgpool", [x]), 192, 1, 1)
        x = g.add("concat", [b1, b7, bd, bp])
    b3 = g.conv_bn(g.conv_bn(x, 192, 1, 1), 320, 3, 3, "valid", 2)      # mixed 8: 8 x 8
    b7 = g.conv_bn(g.conv_bn(g.conv_bn(g.conv_bn(x, 192, 1, 1), 192, 1, 7), 192, 7, 1), 192, 3, 3, "valid", 2)
    x = g.add("concat", [b3, b7, g.add("maxpool", [x])])
    for _ in range(2):                                                  # mixed 9, 10
        b1 = g.conv_bn(x, 320, 1, 1)
        b3 = g.conv_bn(x, 384, 1, 1)
        b3 = g.add("concat", [g.conv_bn(b3, 384, 1, 3), g.conv_bn(b3, 384, 3, 1)])
        bd = g.conv_bn(g.conv_bn(x, 448, 1, 1), 384, 3, 3)
        bd = g.add("concat", [g.conv_bn(bd, 384, 1, 3), g.conv_bn(bd, 384, 3, 1)])
        bp = g.conv_bn(g.add("avgpool", [x]), 192, 1, 1)
        x = g.add("concat", [b1, b3, bd, bp])
    g.add("gap", [x])
    return g.layers


def keras_layer_order(layers):
    """`model.layers` order of a Keras functional model [TF-2.1 network.py: _map_graph_network]: layers sorted by depth
    (longest path to the output, deepest first) and inside one depth by the index of a depth-first walk from the output that
    numbers a layer when it is first reached and visits a layer's inputs in call order."""
    by_name = {l[0]: l for l in layers}
    index, depth = {}, {}

    def walk(name):                                             # (iterative: the graph is ~300 layers deep)
        stack = [(name, 0)]
        while stack:
            n, k = stack.pop()
            if k == 0:
                if n in index:
                    continue
                index[n] = len(index)
            ins = by_name[n][2]
            if k < len(ins):
                stack.append((n, k + 1))
                stack.append((ins[k], 0))
    out = layers[-1][0]
    walk(out)
    depth[out] = 0
    # longest distance to the output: relax consumers before producers (reverse creation order is topological)
    for name, _, ins, _ in reversed(layers):
        d = depth.setdefault(name, 0)
        for i in ins:
            depth[i] = max(depth.get(i, 0), d + 1)
    return sorted((l for l in layers if l[0] in index), key=lambda l: (-depth[l[0]], index[l[0]]))


class InceptionV3(Net):
    """InceptionV3 feature network (include_top=False, pooling="avg"): (N, H, W, 3) in [-1, 1] -> (N, 2048)."""

    def __init__(self, rng=None):
        super().__init__()
        rng = rng or np.random.default_rng(0)
        self.graph = inception_v3_graph()
        cin = {}
        shapes = {}
        for name, kind, ins, p in self.graph:                  # channel bookkeeping in creation order
            if kind == "input":
                cin[name] = 3
            elif kind == "conv":
                shapes[name] = (p["kernel"][0], p["kernel"][1], cin[ins[0]], p["filters"])
                cin[name] = p["filters"]
            elif kind == "concat":
                cin[name] = sum(cin[i] for i in ins)
            else:
                cin[name] = cin[ins[0]]
        self._slots = {}
        for name, kind, ins, p in keras_layer_order(self.graph):
            if kind == "conv":
                kh, kw, ci, co = shapes[name]
                std = np.sqrt(2.0 / (kh * kw * ci))
                self._slots[name] = self.add_weight(name + "_kernel", (rng.standard_normal(shapes[name]) * std).astype(np.float32),
                                                    trainable=False)
            elif kind == "bn":
                co = cin[name]
                self._slots[name] = self.add_weight(name + "_beta", np.zeros(co, np.float32), trainable=False)
                self.add_weight(name + "_mean", np.zeros(co, np.float32), trainable=False)
                self.add_weight(name + "_var", np.ones(co, np.float32), trainable=False)
        self.finalize()
        self.pretrained = False
        self._folded = {}

    def load_keras_weights(self, path):
        """np.savez(path, *InceptionV3(include_top=False, weights="imagenet", pooling="avg").get_weights()) from a TensorFlow machine"""
        with np.load(path, allow_pickle=True) as f:
            ws = [f[k] for k in sorted(f.files, key=lambda k: int(k.split("_")[-1]))]
        self.set_weights(ws)
        self.pretrained = True

    def _fold(self, conv_name, bn_name):
        """inference BatchNormalization(scale=False, eps 1e-3) folded into the preceding bias-free convolution"""
        hit = self._folded.get(conv_name)
        if hit is not None and hit[0] == self.epoch:
            return hit[1], hit[2]
        k = self.weights[self._slots[conv_name]]
        beta, mean, var = self.weights[self._slots[bn_name]:self._slots[bn_name] + 3]
        a = torch.rsqrt(var + 1e-3)
        w, b = (k * a).contiguous(), (beta - mean * a).contiguous()
        self._folded[conv_name] = (self.epoch, w, b)
        return w, b

    def __call__(self, x):
        x = self.to_device(x)
        vals = {}
        consumers = {}
        for name, kind, ins, p in self.graph:
            for i in ins:
                consumers[i] = consumers.get(i, 0) + 1
        graph = {l[0]: l for l in self.graph}
        with torch.no_grad():
            for name, kind, ins, p in self.graph:
                if kind == "input":
                    vals[name] = x
                elif kind == "conv":
                    continue                                    # evaluated at its relu (conv -> bn -> relu is one launch)
                elif kind == "bn":
                    continue
                elif kind == "relu":
                    bn = ins[0]
                    conv = graph[bn][2][0]
                    src = graph[conv][2][0]
                    cp = graph[conv][3]
                    w, b = self._fold(conv, bn)
                    spec = ConvSpec(cp["kernel"], stride=cp["stride"], explicit_pad=0 if cp["padding"] == "valid" else None)
                    xin = vals[src]
                    vals[name] = ops.conv_fwd(xin, w, b, spec.geom(tuple(xin.shape), w.shape[-1]), ACT_RELU, 0.0)
                elif kind == "maxpool":
                    vals[name] = ops.maxpool_fwd(vals[ins[0]], 3, 2, 0)
                elif kind == "avgpool":
                    vals[name] = ops.avgpool3_same(vals[ins[0]])
                elif kind == "concat":
                    vals[name] = torch.cat([vals[i] for i in ins], dim=-1)
                elif kind == "gap":
                    v = vals[ins[0]]
                    s1 = ops.nc_reduce(v, None, want_dot=False)[0]
                    vals[name] = s1.reshape(v.shape[0], v.shape[-1]) / float(v.shape[1] * v.shape[2])
            return vals[self.graph[-1][0]]


def preprocess_input(images):
    """keras.applications.inception_v3.preprocess_input (mode "tf"): x / 127.5 - 1"""
    return np.asarray(images, dtype=np.float32) / 127.5 - 1.0


class InceptionFeatureExtractor:
    """inception_distance.py:9-28."""

    def __init__(self, input_shape, weights_path=None, seed=0):
        self.input_shape = tuple(input_shape)
        self.model = InceptionV3(rng=np.random.default_rng(seed))
        if weights_path is not None:
            self.model.load_keras_weights(weights_path)
        else:
            import sys
            print("InceptionFeatureExtractor: no imagenet weights file given -- seeded random weights; FID / KID values "
                  "are only comparable between runs of this code", file=sys.stderr)

    def get_features(self, images, max_chunk_size=1000, batch_size=32):
        n_imgs = images.shape[0]
        features = np.zeros((n_imgs, 2048), np.float32)
        n_chunks = 1 + n_imgs // max_chunk_size
        for i in range(n_chunks):
            chunk_begin = i * max_chunk_size
            chunk_end = min((i + 1) * max_chunk_size, n_imgs)
            if chunk_end - chunk_begin <= 0:
                break
            pre = preprocess_input(images[chunk_begin:chunk_end])
            for s in range(0, pre.shape[0], batch_size):            # keras predict: batches of 32
                features[chunk_begin + s:chunk_begin + s + batch_size] = self.model(pre[s:s + batch_size]).cpu().numpy()
        return features


def compute_FID(features_g, features_r):
    """inception_distance.py:30-45."""
    import scipy.linalg
    mean_g, mean_r = np.mean(features_g, axis=0), np.mean(features_r, axis=0)
    cov_g, cov_r = np.cov(features_g, rowvar=False), np.cov(features_r, rowvar=False)
    centroid_distance = np.linalg.norm(mean_g - mean_r) ** 2
    covariance_distance = np.trace(cov_g + cov_r - 2 * scipy.linalg.sqrtm(np.dot(cov_g, cov_r)))
    return centroid_distance + np.real(covariance_distance)


def _polynomial_kernel(x, y, degree=3, coef0=1.0):
    """sklearn.metrics.pairwise.polynomial_kernel with gamma=None: (x.y / n_features + coef0) ** degree"""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    return (x @ y.T / x.shape[1] + coef0) ** degree


def compute_KID(features_g, features_r):
    """inception_distance.py:47-59 (eq. 4 of arXiv:1801.01401)."""
    k_gg = _polynomial_kernel(features_g, features_g)
    k_rr = _polynomial_kernel(features_r, features_r)
    k_gr = _polynomial_kernel(features_g, features_r)
    m, n = features_g.shape[0], features_r.shape[0]
    term1 = (1 / (m * (m - 1))) * (np.sum(k_gg) - np.sum(np.diagonal(k_gg)))
    term2 = (1 / (n * (n - 1))) * (np.sum(k_rr) - np.sum(np.diagonal(k_rr)))
    term3 = (1 / (m * n)) * np.sum(k_gr)
    return term1 + term2 - 2 * term3
