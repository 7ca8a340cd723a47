"""LatentGAN (reference: confignet/latent_gan.py): MLP generator/discriminator on encoder latents."""
import json
import os

import numpy as np
import torch

from . import ops, optim
from .confignet_utils import merge_configs
from .dnn_models.building_blocks import MLPSimple
from .losses import GAN_D_loss, GAN_G_loss, gradient_regularization
from .losses import total as total_loss
from .nn import backward_into_arenas

DEFAULT_CONFIG = {
    "latent_dim": None,
    "optimizer": {"lr": 0.00005, "beta_1": 0.0, "beta_2": 0.9, "amsgrad": False},
    "batch_size": 32,
    "num_mlp_layers": 3,
    "latent_distribution_type": "normal",
    "hidden_layer_size_multiplier": 1.5,
    "n_samples_for_metrics": 1000,
    "verbose_log_period": 500,
    "logging_img_square_size": 6,
    # Not in the reference: draw the input latents on the GPU (torch Philox stream) instead of np.random.  The
    # reference's np.random.normal costs ~29 ms per (4096,145) batch on the host -- 20x the device time of a step.
    "device_latent_sampling": False,
}


class LatentGAN:
    def __init__(self, config, seed=None):
        self.config = merge_configs(DEFAULT_CONFIG, config)
        self._rng = np.random.default_rng(seed)
        self.generator = self.generator_smoothed = self.discriminator = None
        self.initialize_network()

    @classmethod
    def load(cls, file_path):
        with open(file_path, "r") as fp:
            config = json.load(fp)
        gan = cls(config)
        gan.set_weights(np.load(os.path.splitext(file_path)[0] + ".npz", allow_pickle=True))
        return gan

    def save(self, output_dir, output_filename):
        os.makedirs(output_dir, exist_ok=True)
        np.savez(os.path.join(output_dir, output_filename + ".npz"), **self.get_weights())
        with open(os.path.join(output_dir, output_filename + ".json"), "w") as fp:
            json.dump(self.config, fp, indent=4)

    def get_weights(self):
        out = {}
        for key, net in (("generator_weights", self.generator), ("smoothed_generator_weights", self.generator_smoothed),
                         ("discriminator_weights", self.discriminator)):
            lst = net.get_weights()
            arr = np.empty(len(lst), dtype=object)
            arr[:] = lst
            out[key] = arr
        return out

    def set_weights(self, weights):
        self.generator.set_weights(weights["generator_weights"])
        self.generator_smoothed.set_weights(weights["smoothed_generator_weights"])
        self.discriminator.set_weights(weights["discriminator_weights"])

    def initialize_network(self):
        L = self.config["latent_dim"]
        hidden = int(L * self.config["hidden_layer_size_multiplier"])
        n = self.config["num_mlp_layers"]
        self.generator = MLPSimple(n, L, hidden, L, rng=self._rng)
        self.generator_smoothed = MLPSimple(n, L, hidden, L, rng=self._rng)
        self.generator_smoothed.copy_weights_from(self.generator)
        self.discriminator = MLPSimple(n, L, hidden, 1, rng=self._rng)

    latent_log = None      # a list: receives every device-drawn latent batch (host copy) while a checker replays the draws

    def sample_input_latent_vector(self, n_samples):
        """latent_gan.py:88-95.  config["device_latent_sampling"] (not in the reference; the FAST path of BASELINE.json configs[4]:
        2.4 ms instead of 76 ms per step at batch 4096, of which 58 ms are the host's two np.random.normal draws) takes the same
        distribution from torch's device generator; the default keeps the reference's np.random stream."""
        if self.config.get("device_latent_sampling"):
            shape = (n_samples, self.config["latent_dim"])
            if self.config["latent_distribution_type"] == "uniform":
                z = torch.rand(shape, device=self.generator.device) * 2.0 - 1.0
            else:
                z = torch.randn(shape, device=self.generator.device)
            if self.latent_log is not None:
                self.latent_log.append(z.detach().cpu().numpy())
            return z
        if self.config["latent_distribution_type"] == "uniform":
            return np.random.uniform(-1, 1, (n_samples, self.config["latent_dim"]))
        return np.random.normal(0, 1, (n_samples, self.config["latent_dim"]))

    def _discriminator_loss(self, real_embeddings, fake_embeddings):
        real = real_embeddings.detach().requires_grad_(True)
        out_real = self.discriminator(real, twice_differentiable=True)
        out_fake = self.discriminator(fake_embeddings.detach())
        losses = {"GAN_loss_real": GAN_D_loss(1.0, out_real), "GAN_loss_fake": GAN_D_loss(0.0, out_fake),
                  "gp_loss": gradient_regularization(out_real, real)}
        losses["loss_sum"] = total_loss(losses.values())
        return losses

    def discriminator_training_step(self, gt_embeddings, optimizer):
        """latent_gan.py:117-149."""
        bs = self.config["batch_size"]
        latents = self.sample_input_latent_vector(bs)
        with torch.no_grad():
            fake = self.generator(latents)
        idx = np.random.randint(0, gt_embeddings.shape[0], bs)
        real = self.discriminator.to_device(gt_embeddings[idx] if not torch.is_tensor(gt_embeddings)
                                            else gt_embeddings[torch.as_tensor(idx, device=gt_embeddings.device)])
        losses = self._discriminator_loss(real, fake)
        backward_into_arenas(losses["loss_sum"], [self.discriminator])
        optimizer.apply_gradients(self.discriminator)
        return losses

    def generator_training_step(self, optimizer):
        """latent_gan.py:151-165."""
        latents = self.sample_input_latent_vector(self.config["batch_size"])
        self.discriminator.requires_grad_(False)
        try:
            losses = {"gan_loss": GAN_G_loss(self.discriminator(self.generator(latents)))}
            losses["loss_sum"] = total_loss(losses.values())
            backward_into_arenas(losses["loss_sum"], [self.generator])
        finally:
            self.discriminator.requires_grad_(True)
        optimizer.apply_gradients(self.generator)
        return losses

    def update_smoothed_weights(self, smoother_alpha=0.999):
        ops.ema_step(self.generator_smoothed.arena, self.generator.arena, smoother_alpha)
        self.generator_smoothed.mark_updated()

    def extract_embeddings(self, confignet_model, training_set, max_chunk_size=1000):
        """latent_gan.py:218-232."""
        n_imgs = training_set.imgs.shape[0]
        embeddings = np.zeros((n_imgs, self.config["latent_dim"]), np.float32)
        for s in range(0, n_imgs, max_chunk_size):
            embeddings[s:s + max_chunk_size], _ = confignet_model.encode_images(training_set.imgs[s:s + max_chunk_size])
        return embeddings

    def train(self, training_set, confignet_model, output_dir, log_dir, n_iters):
        """latent_gan.py:234-247; of write_logs (l.176-200) the periodic checkpoint is kept (TensorBoard images and FID/KID
        are out of scope)."""
        from . import parallel
        gt_embeddings = self.extract_embeddings(confignet_model, training_set)
        parallel.broadcast_weights([self.generator, self.generator_smoothed, self.discriminator])
        optimizer = optim.Adam(**self.config["optimizer"])
        for step_number in range(n_iters):
            d_loss = self.discriminator_training_step(gt_embeddings, optimizer)
            g_loss = self.generator_training_step(optimizer)
            self.update_smoothed_weights()
            print("[step: %d] [D loss: %f] [G loss: %f]" % (step_number, d_loss["loss_sum"], g_loss["loss_sum"]))
            if output_dir is not None and step_number % self.config["verbose_log_period"] == 0 and parallel.rank() == 0:
                self.save(os.path.join(output_dir, "checkpoints"), str(step_number).zfill(6))

    def generate_latents(self, n_samples, truncation=1.0):
        z = self.sample_input_latent_vector(n_samples) * truncation
        return self.generator_smoothed.predict(z.cpu().numpy() if torch.is_tensor(z) else z)
