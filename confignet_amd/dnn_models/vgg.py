"""VGG-19 / VGG-16 convolutional slices up to block4_conv2 (keras.applications definitions used by
confignet/perceptual_loss.py:19-41) on HIP kernels.  The weights are frozen (no filter gradient).
Offline there are no imagenet / VGGFace weights: random He-normal stand-ins are used and can be
replaced through set_weights() with the keras `get_weights()` list of the sliced model."""
import numpy as np
import torch

from .. import functional as F
from ..nn import Net, he_normal
from ..ops import ACT_RELU, ConvSpec

VGG19_CFG = [64, 64, "P", 128, 128, "P", 256, 256, 256, 256, "P", 512, 512]
VGG16_CFG = [64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512]
VGG19_TAPS = (0, 1, 5, 9)    # conv ordinals of keras layers[1, 2, 8, 13]
VGG16_TAPS = (0, 1, 5, 8)    # conv ordinals of keras layers[1, 2, 8, 12]
C3 = ConvSpec((3, 3))


class VGGFeatures(Net):
    def __init__(self, cfg, taps, rng=None):
        super().__init__()
        rng = rng or np.random.default_rng(1234)
        self.cfg, self.taps = cfg, taps
        cin = 3
        for i, item in enumerate(c for c in cfg if c != "P"):
            self.add_weight("conv%d/kernel" % i, he_normal(rng, (3, 3, cin, item)), trainable=False)
            self.add_weight("conv%d/bias" % i, np.zeros(item, np.float32), trainable=False)
            cin = item
        self.finalize()

    def __call__(self, x_pre):
        feats, ci, x = [], 0, x_pre
        for item in self.cfg:
            if item == "P":
                x = F.maxpool(x, 2, 2)
            else:
                x = F.conv(x, self.weights[2 * ci], self.weights[2 * ci + 1], C3, ACT_RELU)
                if ci in self.taps:
                    feats.append(x)
                ci += 1
        return feats
