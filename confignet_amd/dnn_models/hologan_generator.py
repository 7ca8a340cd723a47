"""HologanGenerator (reference: confignet/dnn_models/hologan_generator.py) on HIP kernels."""
import numpy as np
import torch

from .. import functional as F
from ..nn import Net, glorot_uniform
from ..ops import ACT_LRELU, ACT_TANH, ConvSpec
from .building_blocks import KERAS_LRELU, TF_LRELU, conv_adain

C3_UP = ConvSpec((3, 3, 3), up=1)
C3 = ConvSpec((3, 3, 3))
C1 = ConvSpec((1, 1))
C4 = ConvSpec((4, 4))
C4_UP = ConvSpec((4, 4), up=1)


class HologanGenerator(Net):
    def __init__(self, latent_dim, output_shape, n_adain_mlp_units, n_adain_mlp_layers, gen_output_activation,
                 rng=None):
        super().__init__()
        assert n_adain_mlp_layers == 2, "the reference always builds 2-layer AdaIN MLPs"
        assert gen_output_activation == "tanh"
        rng = rng or np.random.default_rng()
        self.latent_dim = latent_dim
        self.output_img_shape = tuple(output_shape)
        self.const_shape = (4, 4, 4, 512)
        res = self.output_img_shape[0]
        assert res in (128, 256, 512), "generator emits 128/256/512 only (hologan_generator.py:159-170)"
        u = n_adain_mlp_units

        def conv(name, shape):
            self.add_weight(name + "/kernel", glorot_uniform(rng, shape))
            self.add_weight(name + "/bias", np.zeros(shape[-1], np.float32))

        def conv_adain_w(name, shape):
            conv(name, shape)
            c = shape[-1]
            for i, s in enumerate([(latent_dim, u), (u, 2 * c)]):
                self.add_weight("%s/adain_mlp%d/kernel" % (name, i), glorot_uniform(rng, s))
                self.add_weight("%s/adain_mlp%d/bias" % (name, i), np.zeros(s[1], np.float32))

        # learned_input: Dense(32768, kernel zeros, bias ones) (l.24-27)
        self.add_weight("learned_input/kernel", np.zeros((1, 32768), np.float32))
        self.add_weight("learned_input/bias", np.ones(32768, np.float32))
        conv_adain_w("map_3d_0", (3, 3, 3, 512, 256))
        conv_adain_w("map_3d_1", (3, 3, 3, 256, 128))
        conv("map_3d_post_0", (3, 3, 3, 128, 64))
        conv("map_3d_post_1", (3, 3, 3, 64, 64))
        conv("projection_conv", (1, 1, 1024, 512))
        conv_adain_w("map_2d_0", (4, 4, 512, 256))
        conv_adain_w("map_2d_1", (4, 4, 256, 64))
        conv_adain_w("map_2d_2", (4, 4, 64, 32))
        last = 32
        self.n_2d = 3
        if res > 128:
            conv_adain_w("map_2d_2b", (4, 4, 32, 32))
            self.n_2d += 1
        if res > 256:
            conv_adain_w("map_2d_2c", (4, 4, 32, 16))
            self.n_2d += 1
            last = 16
        conv("map_final", (4, 4, last, 3))
        self.finalize()

    mlp_bank = True            # (False: one launch per Dense layer -- cross-check)

    def build_input_dict(self, latent_vector, rotation):
        """hologan_generator.py:109-127."""
        d = {}
        zs = latent_vector if isinstance(latent_vector, list) else [latent_vector] * 5
        for k, z in zip(("z_3d_0", "z_3d_1", "z_2d_0", "z_2d_1", "z_2d_2"), zs):
            d[k] = z
        d["rotation"] = rotation
        return d

    def __call__(self, inputs):
        if not isinstance(inputs, dict):
            inputs = self.build_input_dict(inputs[0], inputs[1])
        inp = {k: self.to_device(v) for k, v in inputs.items()}
        w = self.weights
        n = inp["z_3d_0"].shape[0]
        # learned constant: zeros(N,1) @ kernel + bias (l.133-136)
        zeros = torch.zeros((n, 1), device=self.device, dtype=torch.float32)
        x = F.linear(zeros, w[0], w[1]).reshape(n, *self.const_shape)
        zs = [inp["z_2d_0"], inp["z_2d_1"], inp["z_2d_2"], inp["z_2d_2"], inp["z_2d_2"]]
        sbs = [None] * (2 + self.n_2d)
        if self.mlp_bank and all(z.dtype == torch.float32 and z.dim() == 2 and z.shape[0] <= 32 for z in [inp["z_3d_0"], inp["z_3d_1"]] + zs):
            # the AdaIN MLPs of every layer (l.119-124 of the reference build them per layer; they only read the latents): one
            # grouped launch per MLP layer for the whole pass
            sets = [w[4:8], w[10:14]] + [w[20 + 6 * j + 2:20 + 6 * j + 6] for j in range(self.n_2d)]
            sbs = F.mlp_bank([inp["z_3d_0"], inp["z_3d_1"]] + zs[:self.n_2d], sets, TF_LRELU)
        x = conv_adain(x, inp["z_3d_0"], w[2:8], C3_UP, sbs[0])       # UpSampling3D folded (l.139-142)
        x = conv_adain(x, inp["z_3d_1"], w[8:14], C3_UP, sbs[1])      # l.143-144
        x = F.rotate3d(x, inp["rotation"])                            # l.147-148
        x = F.conv(x, w[14], w[15], C3, ACT_LRELU, KERAS_LRELU)       # map_3d_post (l.49-54,151)
        x = F.conv(x, w[16], w[17], C3, ACT_LRELU, KERAS_LRELU)
        s = x.shape
        x = x.reshape(s[0], s[1], s[2], s[3] * s[4])                  # depth collapse (l.153-156)
        x = F.conv(x, w[18], w[19], C1, ACT_LRELU, TF_LRELU)          # projection_conv (l.56,157)
        i = 20
        for j in range(self.n_2d):
            x = conv_adain(x, zs[j], w[i:i + 6], C4 if j == 0 else C4_UP, sbs[2 + j])   # UpSampling2D folded (l.159-170)
            i += 6
        return F.conv(x, w[i], w[i + 1], C4_UP, ACT_TANH)             # map_final (l.101,172)

    def predict(self, inputs, batch_size=32):
        if not isinstance(inputs, dict):
            inputs = self.build_input_dict(inputs[0], inputs[1])
        n = len(inputs["rotation"])
        outs = []
        with torch.no_grad():
            for s in range(0, n, batch_size):
                chunk = {k: np.asarray(v[s:s + batch_size], dtype=np.float32) if not torch.is_tensor(v) else v[s:s + batch_size]
                         for k, v in inputs.items()}
                outs.append(self(chunk).cpu().numpy())
        return np.concatenate(outs, axis=0)
