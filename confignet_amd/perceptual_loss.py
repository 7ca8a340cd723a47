"""PerceptualLoss (reference: confignet/perceptual_loss.py) on HIP kernels."""
import torch

from . import functional as F
from .dnn_models.vgg import VGG16_CFG, VGG16_TAPS, VGG19_CFG, VGG19_TAPS, VGGFeatures, VGGLossFn


class PerceptualLoss:
    def __init__(self, input_shape, model_type="imagenet", rng=None):
        self.input_shape, self.model_type = input_shape, model_type
        if model_type == "VGGFace":
            self._pretrained_dnn_activations = VGGFeatures(VGG16_CFG, VGG16_TAPS, rng)
        elif model_type == "imagenet":
            self._pretrained_dnn_activations = VGGFeatures(VGG19_CFG, VGG19_TAPS, rng)
        else:
            raise ValueError(model_type)

    fused_tape = True          # (False: one tape node per layer and per loss term -- cross-check)

    def _preprocess_input(self, img):
        return F.vggface_preprocess(img) if self.model_type == "VGGFace" else F.caffe_preprocess(img)

    def _activations(self, img, need_grad):
        net = self._pretrained_dnn_activations
        img = net.to_device(img)
        if img.dim() == 3:
            img = img.unsqueeze(0)
        if need_grad:
            return net(self._preprocess_input(img))
        with torch.no_grad():
            return net(self._preprocess_input(img))

    def features(self, img):
        """phi(img) at the tapped layers, without a tape: for a side of the loss that does not change between calls
        (the target image of the fine-tune loop), computed once and passed back as `cached`."""
        return [f.detach() for f in self._activations(img, False)]

    def loss(self, predicted, data, cached=None):
        """sum over the 4 tapped layers of mean((phi(predicted)-phi(data))^2) (perceptual_loss.py:43-82).
        Symmetric; the side that carries gradient (the generated image) keeps its activations.  `cached`: the
        activations of the constant side from `features()` (same values as recomputing them, as the reference does)."""
        p_grad = torch.is_tensor(predicted) and predicted.requires_grad
        d_grad = torch.is_tensor(data) and data.requires_grad
        if p_grad and not d_grad:
            live, const = predicted, data
        else:
            live, const = data, predicted
        live_grad = torch.is_tensor(live) and live.requires_grad
        const_grad = torch.is_tensor(const) and const.requires_grad
        if self.fused_tape and live_grad and not const_grad and torch.is_grad_enabled():
            # the whole stack + the four terms as one tape node (VGGLossFn: one elementwise pass per layer in the backward pass)
            fc = cached if cached is not None else self.features(const)
            net = self._pretrained_dnn_activations
            img = net.to_device(live)
            if img.dim() == 3:
                img = img.unsqueeze(0)
            return VGGLossFn.apply(net, self._preprocess_input(img), None, *[f.detach() for f in fc])
        fl = self._activations(live, live_grad)
        fc = cached if cached is not None else self._activations(const, const_grad)
        total = 0
        for a, b in zip(fl, fc):
            total = total + F.mse_sum(a, b)
        return total

    def loss_groups(self, predicted, const_features, sizes):
        """The perceptual term of consecutive groups of samples of ONE stacked batch as separate scalars: `predicted` (the
        generated images of all groups, carrying gradient) goes through the feature stack once, `const_features` are the
        activations of the stacked ground-truth images from features().  Returns a (G,) tensor; element g equals
        loss(predicted[group g], data[group g])."""
        if self.fused_tape and torch.is_tensor(predicted) and predicted.requires_grad and torch.is_grad_enabled():
            net = self._pretrained_dnn_activations
            return VGGLossFn.apply(net, self._preprocess_input(net.to_device(predicted)), tuple(int(s) for s in sizes),
                                   *[f.detach() for f in const_features])
        fl = self._activations(predicted, True)
        total = 0
        for a, b in zip(fl, const_features):
            total = total + F.mse_group_sums(a, b, sizes)
        return total
