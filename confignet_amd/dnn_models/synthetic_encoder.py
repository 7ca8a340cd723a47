"""SyntheticDataEncoder (reference: confignet/dnn_models/synthetic_encoder.py) on HIP kernels."""
from collections import OrderedDict

import numpy as np
import torch

from ..nn import Net, glorot_uniform
from .building_blocks import KERAS_LRELU, mlp_forward


class _InputMLP:
    """View of one per-input MLP (the reference exposes .per_facemodel_input_mlps[name].predict)."""

    def __init__(self, owner, first, num_in, num_out):
        self.owner, self.first, self.num_in, self.num_out = owner, first, num_in, num_out

    def __call__(self, x):
        return mlp_forward(self.owner.to_device(x), self.owner.weights[self.first:self.first + 4], KERAS_LRELU)

    def predict(self, x, batch_size=32):
        with torch.no_grad():
            return self(np.asarray(x, dtype=np.float32)).cpu().numpy()


class SyntheticDataEncoder(Net):
    def __init__(self, synthetic_encoder_inputs, num_layers, rng=None):
        super().__init__()
        assert isinstance(synthetic_encoder_inputs, OrderedDict)       # synthetic_encoder.py:14
        assert num_layers == 2
        rng = rng or np.random.default_rng()
        self.facemodel_param_names = list(synthetic_encoder_inputs.keys())
        self.per_facemodel_input_mlps = {}
        for name in self.facemodel_param_names:
            din, dout = synthetic_encoder_inputs[name][0], synthetic_encoder_inputs[name][1]
            first = len(self._entries)
            for i, s in enumerate([(din, din), (din, dout)]):
                self.add_weight("mlp_%s/dense%d/kernel" % (name, i), glorot_uniform(rng, s))
                self.add_weight("mlp_%s/dense%d/bias" % (name, i), np.zeros(s[1], np.float32))
            self.per_facemodel_input_mlps[name] = _InputMLP(self, first, din, dout)
        self.finalize()

    def build_input_dictionary(self, inputs):
        if isinstance(inputs, list):
            return dict(zip(self.facemodel_param_names, inputs))
        d, used = {}, 0
        for name in self.facemodel_param_names:
            n_in = self.per_facemodel_input_mlps[name].num_in
            d[name] = inputs[:, used:used + n_in]
            used += n_in
        return d

    def __call__(self, inputs):
        if not isinstance(inputs, dict):
            inputs = self.build_input_dictionary(inputs)
        outs = [self.per_facemodel_input_mlps[n](inputs[n]) for n in self.facemodel_param_names]
        return torch.cat(outs, dim=1)

    def predict(self, inputs, batch_size=32):
        with torch.no_grad():
            return self(inputs).cpu().numpy()
