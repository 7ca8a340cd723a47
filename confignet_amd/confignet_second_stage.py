"""ConfigNet second stage (reference: confignet/confignet_second_stage.py): adds the ResNet-50 real
encoder trained end-to-end, the normalised latent-regression loss and one-shot fine-tuning."""
import time
import os

import numpy as np
import torch

from . import confignet_utils, ops, optim, parallel
from .confignet_first_stage import DEFAULT_CONFIG, ConfigNetFirstStage, frozen
from .dnn_models.hologan_generator import HologanGenerator
from .dnn_models.real_encoder import RealEncoder
from .losses import GAN_D_loss, GAN_G_loss, GAN_G_losses, compute_latent_discriminator_loss, eye_loss, mean_squared_error, normalized_latent_regression
from .losses import total as total_loss
from .nn import Net, backward_into_arenas
from .perceptual_loss import PerceptualLoss


class _Variables(Net):
    """A handful of free tensors optimised next to a network (the tf.Variables of fine_tune_on_img)."""

    def __init__(self, arrays):
        super().__init__()
        for i, a in enumerate(arrays):
            self.add_weight("var%d" % i, a)
        self.finalize()


class ConfigNet(ConfigNetFirstStage):
    def __init__(self, config, initialize=True, seed=None):
        self.config = confignet_utils.merge_configs(DEFAULT_CONFIG, config)
        super(ConfigNet, self).__init__(self.config, initialize=False, seed=seed)
        self.config["model_type"] = "ConfigNet"
        self.encoder = None
        self.generator_fine_tuned = None
        self.perceptual_loss_face_reco = PerceptualLoss(self.config["output_shape"], model_type="VGGFace")
        if initialize:
            self.initialize_network()

    def get_weights(self):
        weights = super().get_weights()
        weights["real_encoder_weights"] = self.encoder.get_weights()
        return weights

    def set_weights(self, weights):
        super().set_weights(weights)
        self.encoder.set_weights(weights["real_encoder_weights"])

    def initialize_network(self):
        super(ConfigNet, self).initialize_network()
        self.encoder = RealEncoder(self.config["latent_dim"], self.config["output_shape"],
                                   self.config["rotation_ranges"], rng=self._rng)

    def all_networks(self):
        return super().all_networks() + [self.encoder]

    # ---- training code ----------------------------------------------------------------------------
    def face_reco_loss(self, gt_imgs, gen_imgs, cached=None):
        return self.perceptual_loss_face_reco.loss(gen_imgs, gt_imgs, cached=cached)

    def compute_normalized_latent_regression_loss(self, generator_outputs, labels, deferred=False, regressor_output=None):
        """confignet_second_stage.py:93-107.  The (N, L+3) batch statistics are latent-vector algebra
        (host-side plumbing); the latent regressor itself runs on HIP kernels.

        deferred=True (only _generator_loss passes it: its _generator_update differentiates the term) selects the
        global-batch-statistics form under data parallelism; every other caller -- fine_tune_on_img backpropagates
        loss_sum itself -- gets the ordinary taped term with the statistics of the batch at hand."""
        # regressor_output: the latent regressor's output for these images, already computed (the generator step runs the two
        # halves of the stack on its two streams: the regressor is per-sample up to this point)
        out = self.latent_regressor(generator_outputs) if regressor_output is None else regressor_output
        # config["dp_global_batch_statistics"] (default off): under data parallelism the batch statistics of the GLOBAL batch
        # (one 2 x (L + 3)-float all-reduce forward and one backward per statistic) instead of the per-rank ones
        if deferred and bool(self.config.get("dp_global_batch_statistics", False)) and parallel.active():
            from .losses import DeferredGlobalStatsRegression
            term = DeferredGlobalStatsRegression(out, labels, self.config["latent_regression_weight"])
            self._deferred_terms.append(term)          # differentiated by _generator_update (its collectives run on this thread)
            return term.value
        return normalized_latent_regression(out, labels, self.config["latent_regression_weight"])

    def sample_random_batch_of_images(self, dataset, batch_size=None):
        """confignet_second_stage.py:109-117 (device gather of the staged indices/flips)."""
        self._stage_real("rand", dataset, self.get_batch_size() if batch_size is None else batch_size)
        return self._real_imgs("rand", dataset)

    def _stage_d_batch(self, training_set):
        n = self.get_batch_size()
        self._stage_real("d", training_set, n)
        self._stage("d/enc_idx", np.random.randint(0, training_set.imgs.shape[0], n), torch.int64)

    def _d_fake(self, training_set):
        """confignet_second_stage.py:119-130: fakes are generated from encoded (unflipped) real images."""
        with torch.no_grad():
            input_imgs = ops.gather_images_u8(self._pool(training_set)["imgs"], self._bufs["d/enc_idx"], None)
            latent_vector, rotation = self.encoder(input_imgs)
            return self.generator([latent_vector, rotation])

    def _prestage_key(self, datasets, optimizer):
        """What a batch staged ahead of its step (see _prelaunch_generator_targets) was staged for: the call's arguments, every
        network's weight version (set_weights / load between the two invalidates it) and the static buffers' generation."""
        nets = [n for n in self.all_networks() if n is not self.generator_smoothed]        # (the EMA copy moves after every iteration)
        return (tuple(id(d) for d in datasets), id(optimizer), tuple(n.epoch for n in nets), self._bufs.generation)

    def _stage_ld_batch(self, real_training_set, synth_training_set):
        n = self.get_batch_size()
        self._stage_real("ld", real_training_set, n)
        self._stage_synth("ld", synth_training_set, n)

    def _stage_g_batch(self, real_training_set, synth_training_set, late=None):
        n_synth = self.get_batch_size() // 2
        self._stage_synth("g", synth_training_set, n_synth, late=late)
        self._stage_real("g", real_training_set, self.get_batch_size() - n_synth)

    def latent_discriminator_training_step(self, real_training_set, synth_training_set, optimizer):
        if self._prestaged.pop("ld", None) != self._prestage_key((real_training_set, synth_training_set), optimizer):
            self._stage_ld_batch(real_training_set, synth_training_set)
        else:
            # staged ahead on the step's own stream (_prelaunch_generator_targets): a call that does not replay there (the step on
            # its own, eager dispatch) must not read the buffers before those uploads have run
            self._bufs.wait_staged("ld/")
        self._stagers["ld"] = (real_training_set, synth_training_set, optimizer)

        def device():
            with torch.no_grad():
                real_latents, _ = self.encoder(self._real_imgs("ld", real_training_set))
                params, _, _, _ = self._synth_batch("ld", synth_training_set, imgs=False)
                fake_latents = self.synthetic_encoder(params)
            return self._latent_discriminator_update(real_latents, fake_latents, optimizer)
        return self._run_step("ld", (real_training_set, synth_training_set), optimizer, device)

    def _generator_loss(self, facemodel_params, synth_rotations, synth_imgs, eye_masks, real_imgs, target_features=None):
        """The taped part of ConfigNet.generator_training_step (l.167-211).  target_features: (VGG activations of synth_imgs, of
        real_imgs) computed ahead of the step (_generator_targets), else they are computed here.

        The real-image branch (encoder -> generator -> perceptual / adversarial terms) and the synthetic branch
        (synthetic encoder -> generator -> perceptual / eye / adversarial terms) only meet in the latent regressor
        and in the loss sum, so the real branch is issued on a second stream: forward here, and backward too, since
        autograd replays every node on the stream its forward ran on.  Captured into the step's HIP graph the two
        branches become parallel paths, and the many small launches of one overlap the big ones of the other."""
        if self.merge_generator_passes:
            return self._generator_loss_merged(facemodel_params, synth_rotations, synth_imgs, eye_masks, real_imgs)
        cfg = self.config
        n_synth, n_real = synth_imgs.shape[0], real_imgs.shape[0]
        losses = {}
        main = torch.cuda.current_stream()
        side = self._branch_stream if self.fork_generator_step else main
        if side is not main:
            side.wait_stream(main)
        from .graphs import segment_break
        with torch.cuda.stream(side):
            real_latents, real_rotations = self.encoder(real_imgs)
            self._g_cut = ([real_latents, real_rotations], [self.encoder])     # the encoder hangs on the tape by these two only
            generator_output_real = self.generator((real_latents, real_rotations))
            image_loss_real = cfg["image_loss_weight"] * self.perceptual_loss.loss(
                real_imgs, generator_output_real, cached=None if target_features is None else target_features[1])
        synth_latents = self.synthetic_encoder(facemodel_params)
        generator_output_synth = self.generator((synth_latents, synth_rotations))
        losses["image_loss_synth"] = cfg["image_loss_weight"] * self.perceptual_loss.loss(
            synth_imgs, generator_output_synth, cached=None if target_features is None else target_features[0])
        losses["image_loss_real"] = None                      # (keeps the reference's key order; filled after the join)
        losses["eye_loss"] = cfg["eye_loss_weight"] * eye_loss(synth_imgs, generator_output_synth, eye_masks)
        # Everything above -- encoder, both generator passes, the four VGG-19 passes: the heavy MFMA-bound half of the step's
        # forward -- reads no discriminator weight, so in graph mode the iteration replays it NEXT TO the three discriminator-
        # type steps (run_concurrently(..., then=)); below this point the step needs their updated weights.
        if side is not main:
            main.wait_stream(side)
        segment_break(early=True)
        if side is not main:
            side.wait_stream(main)
        # The latent regressor is per-sample (instance normalisation) up to its output: the two halves of the reference's stacked
        # batch (l.188-197) go through it on the two streams, each right behind its own branch's heads, instead of one pass over
        # the concatenation after the join -- the single-stream stretch of the step's tail (regressor forward, and its backward
        # before the branches' backward passes can start) is halved, and the pixel-sized torch.cat is gone.
        split_lr = cfg["latent_regression_weight"] > 0.0 and self.split_latent_regressor
        reg_real = reg_synth = None
        with torch.cuda.stream(side):
            gan_real = GAN_G_losses(self.discriminator(generator_output_real).values())
            out_real = self.latent_discriminator(real_latents)
            if split_lr and not self.regressor_real_half_on_main:
                reg_real = self.latent_regressor(generator_output_real)
        for i, l in enumerate(GAN_G_losses(self.synth_discriminator(generator_output_synth).values())):
            losses["GAN_loss_synth_" + str(i)] = l
        out_synth = self.latent_discriminator(synth_latents)
        if split_lr:
            reg_synth = self.latent_regressor(generator_output_synth)
        if side is not main:
            main.wait_stream(side)                            # join: everything below needs both branches
            if not torch.cuda.is_current_stream_capturing():
                # eager dispatch: blocks allocated on the side stream are consumed on the main stream from here on -- tell the
                # caching allocator, which otherwise hands them back to the side stream's pool on free (wait_stream orders the
                # kernels, not the allocator).  Captured graphs allocate from their private pool.
                for t in [real_latents, real_rotations, generator_output_real, image_loss_real, out_real] + gan_real + ([reg_real] if reg_real is not None else []):
                    t.record_stream(main)
        if split_lr and reg_real is None:
            # the REAL half's regressor pass on the calling stream as well: the real branch (encoder -> generator -> VGG -> heads) is
            # the longer chain of the step by the whole ResNet-50 forward and backward; the synthetic branch's stream has that much
            # slack, and the regressor's backward then runs there next to the real branch's VGG / discriminator backward
            reg_real = self.latent_regressor(generator_output_real)
            # (the discriminator heads of the real half moved the same way: 373.5 against 381.0 images/s -- they stay on the branch)
        losses["image_loss_real"] = image_loss_real
        for i, l in enumerate(gan_real):
            losses["GAN_loss_real_" + str(i)] = l
        # labels: real -> 0, synth -> 1 (l.160-163,195-197); mean over the concatenation
        latent_gan_loss = (n_real * GAN_D_loss(0.0, out_real) + n_synth * GAN_D_loss(1.0, out_synth)) / (n_real + n_synth)
        losses["latent_GAN_loss"] = cfg["domain_adverserial_loss_weight"] * latent_gan_loss
        if cfg["latent_regression_weight"] > 0.0:
            stacked_latents = torch.cat((synth_latents, real_latents), dim=0)
            stacked_rotations = torch.cat((synth_rotations, real_rotations), dim=0)
            labels = torch.cat((stacked_latents, cfg["latent_regressor_rot_weight"] * stacked_rotations), dim=-1)
            if split_lr:
                losses["latent_regression_loss"] = self.compute_normalized_latent_regression_loss(
                    None, labels, deferred=True, regressor_output=torch.cat((reg_synth, reg_real), dim=0))
            else:
                stacked_imgs = torch.cat((generator_output_synth, generator_output_real), dim=0)
                losses["latent_regression_loss"] = self.compute_normalized_latent_regression_loss(stacked_imgs, labels, deferred=True)
        losses["loss_sum"] = total_loss(losses.values())
        return losses

    split_latent_regressor = True
    regressor_real_half_on_main = True

    # OFF by default -- measured (round 4, profiles/round4_schedule_experiments.txt): 478 instead of 537 convolution launches and
    # 1.2 ms less convolution kernel time per iteration, but the step takes 28.9 instead of 26.9 ms and the iteration 49.4 instead
    # of 45.5: in the two-pass form the real branch (encoder -> generator -> VGG -> heads) and the synthetic branch run side by
    # side on two streams from start to end, and that concurrency fills the CUs which the small launches of one chain leave idle;
    # the stacked form has to wait for the encoder before its only generator pass can start.  ConfigNet.merge_generator_passes = True selects it.
    merge_generator_passes = False

    def _generator_loss_merged(self, facemodel_params, synth_rotations, synth_imgs, eye_masks, real_imgs):
        """(See merge_generator_passes: an option, slower end to end.)  The same loss dict from FEWER, LARGER launches: the synthetic and the real half go through the generator as ONE
        stacked batch (synthetic samples first -- the order the latent regressor's stack has anyway, l.188-197), through VGG-19 as
        one stacked batch per side (generated / ground truth), and the perceptual terms of the two halves come out of one
        reduction each (PerceptualLoss.loss_groups).  Every sample sees exactly the arithmetic of the two-pass form: all layers
        are per-sample (AdaIN / instance statistics are per sample; no batch statistics before the latent regression).
        Second stream: the ground-truth VGG pass (no tape, depends on nothing) runs beside encoder + generator, and the
        discriminator heads of the real half beside those of the synthetic half, as before."""
        cfg = self.config
        n_synth, n_real = synth_imgs.shape[0], real_imgs.shape[0]
        losses = {}
        main = torch.cuda.current_stream()
        side = self._branch_stream if self.fork_generator_step else main
        from .graphs import segment_break
        if side is not main:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            gt_features = self.perceptual_loss.features(torch.cat((synth_imgs, real_imgs), dim=0))
        real_latents, real_rotations = self.encoder(real_imgs)
        self._g_cut = ([real_latents, real_rotations], [self.encoder])     # the encoder hangs on the tape by these two only
        synth_latents = self.synthetic_encoder(facemodel_params)
        stacked_latents = torch.cat((synth_latents, real_latents), dim=0)
        stacked_rotations = torch.cat((synth_rotations, real_rotations), dim=0)
        stacked_imgs = self.generator((stacked_latents, stacked_rotations))
        generator_output_synth, generator_output_real = stacked_imgs[:n_synth], stacked_imgs[n_synth:]
        if side is not main:
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                for t in gt_features:
                    t.record_stream(main)
        image_losses = cfg["image_loss_weight"] * self.perceptual_loss.loss_groups(stacked_imgs, gt_features, (n_synth, n_real))
        losses["image_loss_synth"] = image_losses[0]
        losses["image_loss_real"] = image_losses[1]
        losses["eye_loss"] = cfg["eye_loss_weight"] * eye_loss(synth_imgs, generator_output_synth, eye_masks)
        segment_break(early=True)                                      # nothing above reads a discriminator weight
        if side is not main:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            gan_real = GAN_G_losses(self.discriminator(generator_output_real).values())
            out_real = self.latent_discriminator(real_latents)
        for i, l in enumerate(GAN_G_losses(self.synth_discriminator(generator_output_synth).values())):
            losses["GAN_loss_synth_" + str(i)] = l
        out_synth = self.latent_discriminator(synth_latents)
        if side is not main:
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                for t in [out_real] + gan_real:
                    t.record_stream(main)
        for i, l in enumerate(gan_real):
            losses["GAN_loss_real_" + str(i)] = l
        latent_gan_loss = (n_real * GAN_D_loss(0.0, out_real) + n_synth * GAN_D_loss(1.0, out_synth)) / (n_real + n_synth)
        losses["latent_GAN_loss"] = cfg["domain_adverserial_loss_weight"] * latent_gan_loss
        if cfg["latent_regression_weight"] > 0.0:
            labels = torch.cat((stacked_latents, cfg["latent_regressor_rot_weight"] * stacked_rotations), dim=-1)
            losses["latent_regression_loss"] = self.compute_normalized_latent_regression_loss(stacked_imgs, labels, deferred=True)
        losses["loss_sum"] = total_loss(losses.values())
        return losses

    def generator_training_step(self, real_training_set, synth_training_set, optimizer):
        n_synth = self.get_batch_size() // 2
        n_real = self.get_batch_size() - n_synth
        assert n_synth > 0, "the generator step splits the batch into a synthetic and a real half: batch_size >= 2 " \
                            "(the reference's losses are means over an empty synthetic batch, i.e. NaN, at batch_size 1)"
        datasets = (real_training_set, synth_training_set)
        key = self._prestage_key(datasets, optimizer)
        if self._prestaged.pop("g", None) != key:
            self._stage_g_batch(real_training_set, synth_training_set)
        else:
            self._bufs.wait_staged("g/")         # (uploaded on two other streams under the previous generator tail)
        self._stagers["g"] = (real_training_set, synth_training_set, optimizer)
        nets = [self.generator, self.latent_regressor, self.synthetic_encoder, self.encoder]
        self._targets_now = targets = self._generator_targets(datasets, optimizer, key)

        def device():
            targets = self._targets_now          # (read at run time: the step graph keeps the closure of its FIRST call)
            params, synth_rot, synth_imgs, eye_masks = self._synth_batch("g", synth_training_set)
            real_imgs = self._real_imgs("g", real_training_set)
            # (copies: the step keeps the activations for its backward pass, and the next iteration's targets are written while
            # this step's tail is still running)
            feats = None if targets is None else tuple([f.clone() for f in fs] for fs in targets)
            with frozen(self.discriminator, self.synth_discriminator, self.latent_discriminator):
                losses = self._generator_loss(params, synth_rot, synth_imgs, eye_masks, real_imgs, feats)
                self._generator_update(losses, nets, optimizer, cut=self._g_cut)
                self._g_cut = None
            return losses
        out = self._run_step("g", datasets, optimizer, device)
        g = self._graph_of("g", datasets, optimizer)
        if g is not None and targets is not None:
            g.target_graph = self._target_graph_now
        return out

    # The perceptual terms compare the generated images with the VGG-19 activations of the step's GROUND-TRUTH images (l.171-176):
    # two of the step's four VGG passes read nothing but the batch and the frozen VGG weights.  In the pipelined loop (graphs +
    # overlap_discriminators) they are a graph of their own, replayed for iteration t+1 behind the real half of the discriminator
    # step under iteration t's generator tail -- whose last third (generator and encoder backward: chains of small launches) has
    # the chip almost to itself, while the generator step's forward at the start of t+1 competes with three discriminator lines.
    # The batch of t+1 is drawn with it: all four host halves move to the end of iteration t, in the reference's order (D,
    # synth-D, latent-D, G).  prelaunch_generator_targets = False: the four passes inside the step as before.
    prelaunch_generator_targets = True

    def _generator_targets(self, datasets, optimizer, key):
        """The ground-truth VGG activations for the batch staged under "g": None (the step computes them itself) outside the
        pipelined loop; otherwise the outputs of the target graph -- already written if it was replayed ahead for this batch."""
        if not self.use_graphs:
            return None
        real_training_set, synth_training_set = datasets
        g = self._graph_of("g", datasets, optimizer)
        tg = getattr(g, "target_graph", None) if g is not None else None
        # (decided when the step's graph is created and kept for its lifetime: a captured step reads the target graph's outputs)
        if (tg is None) if g is not None else not (self.prelaunch_generator_targets and self.overlap_discriminators and not self.merge_generator_passes):
            return None
        if tg is None:
            from .graphs import StepGraph

            def fn():
                with torch.no_grad():
                    synth = ops.gather_images_u8(self._pool(synth_training_set)["imgs"], self._bufs["g/synth_idx"], None)
                    real = self._real_imgs("g", real_training_set)
                    return (self.perceptual_loss.features(synth), self.perceptual_loss.features(real))
            tg = StepGraph(fn, stream=self._work_stream("main"))
        self._target_graph_now = tg
        ahead, self._targets_ahead = self._targets_ahead, None
        if ahead is not None and ahead[0] == key and ahead[1] is tg:
            torch.cuda.current_stream().wait_stream(ahead[2])       # replayed on that stream under the previous generator tail
            return tg.out
        if ahead is not None:
            torch.cuda.current_stream().wait_stream(ahead[2])       # (for another batch: let it finish before the buffers are rewritten)
        return tg()

    _targets_ahead = None
    _target_graph_now = None
    _targets_now = None

    def _prelaunch_generator_targets(self, pending, follower, ev_early):
        tg = getattr(follower, "target_graph", None)
        st_ld, st_g = self._stagers.get("ld"), self._stagers.get("g")
        ld = next((g for g in pending if getattr(g, "name", None) == "ld"), None)
        ahead = next((g for g in pending if getattr(g, "prelaunched", False)), None)
        if tg is None or tg.graph is None or st_ld is None or st_g is None or ld is None or ahead is None or ev_early is None:
            return
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(ld.stream):                   # the latent-discriminator step's host half (its draws come third)
            self._stage_ld_batch(st_ld[0], st_ld[1])
        with torch.cuda.stream(ahead.stream):
            # the index / flip buffers are read by gathers at the very start of the generator step and by the target graph: free
            # once this iteration's early generator forward has run; everything else is copied behind the tail on the main line
            ahead.stream.wait_event(ev_early)
            self._stage_g_batch(st_g[0], st_g[1], late=cur)
            tg.replay()
        # (keys AFTER the tail's finish(): every network's weight version is the one the next iteration's steps will see)
        self._prestaged["ld"] = self._prestage_key((st_ld[0], st_ld[1]), st_ld[2])
        key = self._prestage_key((st_g[0], st_g[1]), st_g[2])
        self._prestaged["g"] = key
        self._targets_ahead = (key, tg, ahead.stream)

    def training_iteration(self, real_training_set, synth_training_set, discriminator_optimizer, generator_optimizer):
        """One reference training iteration (confignet_second_stage.py:277-288): D, synth-D, latent-D,
        G, EMA.  Returns the four loss dicts (device scalars; no host sync here)."""
        with self._main_line():
            d_steps = [lambda: self.discriminator_training_step(real_training_set, discriminator_optimizer),
                       lambda: self.synth_discriminator_training_step(synth_training_set, discriminator_optimizer),
                       lambda: self.latent_discriminator_training_step(real_training_set, synth_training_set, discriminator_optimizer)]
            g_step = lambda: self.generator_training_step(real_training_set, synth_training_set, generator_optimizer)
            nd, ng = self.config["n_discriminator_updates"], self.config["n_generator_updates"]
            for i in range(nd):
                out = self.run_concurrently(d_steps, then=g_step if (i == nd - 1 and ng >= 1) else None)
                d_loss, synth_d_loss, latent_d_loss = out[:3]
                if len(out) > 3:
                    g_loss = out[3]
            for _ in range(ng - 1 if nd >= 1 else ng):
                g_loss = g_step()
            self.update_smoothed_weights()
        return d_loss, synth_d_loss, latent_d_loss, g_loss

    def setup_training(self, log_dir, synth_training_set, n_samples_for_metrics, attribute_classifier=None,
                       real_training_set=None, validation_set=None):
        """confignet_second_stage.py:255-266 (ControllabilityMetrics: out of scope)."""
        super(ConfigNet, self).setup_training(log_dir, synth_training_set, n_samples_for_metrics, real_training_set)
        if validation_set is None:
            validation_set = real_training_set if real_training_set is not None else synth_training_set
        imgs = validation_set.imgs
        sample_idxs = np.random.randint(0, imgs.shape[0], self.n_checkpoint_samples)
        self._checkpoint_visualization_input["input_images"] = self._host_images(imgs, sample_idxs)
        sample_idxs = np.random.randint(0, imgs.shape[0], n_samples_for_metrics)
        self._generator_input_for_metrics["input_images"] = self._host_images(imgs, sample_idxs)

    @staticmethod
    def _host_images(imgs, idxs):
        """imgs[idxs] as float32 in [-1, 1] on the host (the dataset pool may live in HBM as a uint8 tensor)."""
        sel = imgs[torch.as_tensor(idxs, device=imgs.device)].cpu().numpy() if torch.is_tensor(imgs) else np.asarray(imgs)[idxs]
        return sel.astype(np.float32) / 127.5 - 1.0

    def calculate_metrics(self, output_dir, aml_run=None):
        """confignet_second_stage.py:220-253: FID/KID (first stage) + the perceptual reconstruction metric of the metric
        images through encoder -> smoothed generator; the controllability metrics are out of scope."""
        super(ConfigNet, self).calculate_metrics(output_dir, aml_run)
        inp = self._generator_input_for_metrics["input_images"]
        latents, rotations = self.encode_images(inp)
        generated = self.generator_smoothed.predict(self.generator_smoothed.build_input_dict(latents, rotations))
        vals = []
        with torch.no_grad():
            for s in range(0, len(inp), 16):                               # metric_batch_size 16 (l.231)
                vals.append(float(self.perceptual_loss.loss(inp[s:s + 16], generated[s:s + 16])))
        self.metrics.setdefault("perceptual_loss", []).append(float(np.mean(vals)))
        os.makedirs(output_dir, exist_ok=True)
        np.savetxt(os.path.join(output_dir, "image_metrics.txt"), self.metrics["perceptual_loss"])

    def train(self, real_training_set, synth_training_set, validation_set, attribute_classifier, output_dir, log_dir,
              n_steps=100000, n_samples_for_metrics=1000, aml_run=None):
        """confignet_second_stage.py:268-299."""
        self.setup_training(log_dir, synth_training_set, n_samples_for_metrics, attribute_classifier,
                            real_training_set=real_training_set, validation_set=validation_set)
        parallel.broadcast_weights(self.all_networks())       # data-parallel replicas start from rank 0's weights
        start_step = self.get_training_step_number()
        discriminator_optimizer = optim.Adam(**self.config["optimizer"])
        generator_optimizer = optim.Adam(**self.config["optimizer"])
        # the training loop dispatches through HIP graphs unless config["use_hip_graphs"] says otherwise (direct calls of
        # the step functions stay eager by default: in graph mode the returned loss scalars are the graph's static
        # outputs, overwritten by the next replay)
        self.use_graphs = bool(self.config.get("use_hip_graphs", True))
        self.overlap_discriminators = self.use_graphs and bool(self.config.get("overlap_discriminators", True))
        for _ in range(start_step, n_steps):
            t0 = time.perf_counter()
            d_loss, synth_d_loss, latent_d_loss, g_loss = self.training_iteration(
                real_training_set, synth_training_set, discriminator_optimizer, generator_optimizer)
            torch.cuda.synchronize()
            self.last_iteration_time = time.perf_counter() - t0
            print("[D loss: %f] [synth_D loss: %f] [latent_D_loss: %f] [G loss: %f]" %
                  (d_loss["loss_sum"], synth_d_loss["loss_sum"], latent_d_loss["loss_sum"], g_loss["loss_sum"]))
            confignet_utils.update_loss_dict(self.g_losses, g_loss)
            confignet_utils.update_loss_dict(self.d_losses, d_loss)
            confignet_utils.update_loss_dict(self.synth_d_losses, synth_d_loss)
            confignet_utils.update_loss_dict(self.latent_d_losses, latent_d_loss)
            self.run_checkpoints(output_dir, self.last_iteration_time, aml_run=aml_run)

    # ---- inference ----------------------------------------------------------------------------------
    def encode_images(self, input_images):
        """confignet_second_stage.py:301-308."""
        if not torch.is_tensor(input_images) and input_images.dtype == np.uint8:
            input_images = input_images.astype(np.float32) / 127.5 - 1.0
        return self.encoder.predict(np.asarray(input_images, dtype=np.float32))

    def generate_images(self, latent_vectors, rotations):
        """confignet_second_stage.py:310-319."""
        g = self.generator_fine_tuned if self.generator_fine_tuned is not None else self.generator_smoothed
        return self._generate_images_with(g, latent_vectors, rotations)

    def fine_tune_on_img(self, input_images, n_iters=50, img_output_dir=None, force_neutral_expression=False):
        """confignet_second_stage.py:321-403: Adam(lr 1e-4) on the fine-tuned generator copy, the
        pre/post-expression embeddings (batch mean, tiled), the expression slice and the rotations."""
        if input_images.dtype == np.uint8:
            input_images = (input_images / 127.5) - 1.0
        if len(input_images.shape) == 3:
            input_images = input_images[np.newaxis]
        input_images = np.asarray(input_images, dtype=np.float32)
        emb, rot = self.encoder.predict(input_images)
        if force_neutral_expression:
            n_bs = self.config["facemodel_inputs"]["blendshape_values"][0]
            emb = self.set_facemodel_param_in_latents(emb, "blendshape_values", np.zeros((1, n_bs), np.float32))
        if self.generator_fine_tuned is None:
            self.generator_fine_tuned = HologanGenerator(rng=self._rng, **self._get_generator_kwargs())
        self.generator_fine_tuned.copy_weights_from(self.generator_smoothed)
        gen = self.generator_fine_tuned

        expr_idxs = self.get_facemodel_param_idxs_in_latent("blendshape_values")
        mean_emb = np.mean(emb, axis=0, keepdims=True)
        var = _Variables([mean_emb[:, :expr_idxs[0]], emb[:, expr_idxs], mean_emb[:, expr_idxs[-1] + 1:], rot])
        pre, expr, post, rotations = var.weights
        if force_neutral_expression:
            expr.requires_grad_(False)
        n_imgs = input_images.shape[0]
        imgs_dev = gen.to_device(input_images)
        optimizer = optim.Adam(lr=0.0001)
        w = self.config
        state = {}
        # The target image's VGG19 / VGGFace activations do not change over the loop.  The reference recomputes them
        # every step (l.369-370); here they are computed once -- identical values, 2 of the 6 VGG passes per step saved
        # (set cache_target_features = False for the literal recomputation).
        cache = getattr(self, "cache_target_features", True)
        tgt_vgg = self.perceptual_loss.features(imgs_dev) if cache else None
        tgt_face = self.perceptual_loss_face_reco.features(imgs_dev) if cache else None

        def device_step():
            losses = {}
            with frozen(self.discriminator, self.latent_discriminator, self.latent_regressor):
                pre_t, post_t = pre.repeat(n_imgs, 1), post.repeat(n_imgs, 1)
                embeddings = torch.cat((pre_t, expr, post_t), dim=1)
                out = gen((embeddings, rotations))
                losses["image_loss_real"] = 0.5 * w["image_loss_weight"] * self.perceptual_loss.loss(imgs_dev, out, cached=tgt_vgg)
                losses["face_reco_loss"] = 0.5 * w["image_loss_weight"] * self.face_reco_loss(imgs_dev, out, cached=tgt_face)
                for i, o in enumerate(self.discriminator(out).values()):
                    losses["GAN_loss_real_" + str(i)] = GAN_G_loss(o)
                latent_gan_loss = GAN_D_loss(1.0, self.latent_discriminator(embeddings))
                losses["latent_GAN_loss"] = w["domain_adverserial_loss_weight"] * latent_gan_loss
                labels = torch.cat((embeddings, w["latent_regressor_rot_weight"] * rotations), dim=-1)
                losses["latent_regression_loss"] = self.compute_normalized_latent_regression_loss(out, labels)
                losses["loss_sum"] = total_loss(losses.values())
                backward_into_arenas(losses["loss_sum"], [gen, var])
            # pre/post tiled BEFORE this step's update (what the reference returns after the last step, l.363-364,402)
            state["stale"] = torch.cat((pre_t, expr, post_t), dim=1).detach().clone()
            optimizer.apply_gradients([gen, var], advance=False, slot="ft")
            return {k: v.detach() for k, v in losses.items()}

        # N = 1 makes this loop latency-bound: the step is captured once into a HIP graph and replayed
        if self.use_graphs:
            from .graphs import StepGraph
            # one capture stream for every fine-tune call of this model (a stream per call would leave a deterministic-mode
            # workspace pinned per call: cn_det_ws)
            if getattr(self, "_fine_tune_stream", None) is None:
                self._fine_tune_stream = torch.cuda.Stream()
                self._release_on_exit([self._fine_tune_stream])
            runner = StepGraph(device_step, stream=self._fine_tune_stream)
        else:
            runner = device_step
        log = getattr(self, "fine_tune_loss_log", None)       # a list: receives every step's losses as floats (one sync per step)
        for step_number in range(n_iters):
            optimizer.advance("ft")
            self.last_fine_tune_losses = runner()
            if log is not None:
                log.append({k: float(v) for k, v in self.last_fine_tune_losses.items()})
        stale = state["stale"]
        # the reference returns pre/post tiled before the last optimizer step with the updated expr (l.402)
        result = stale.cpu().numpy()
        result[:, list(expr_idxs)] = expr.detach().cpu().numpy()
        return result, rotations.detach().cpu().numpy()
