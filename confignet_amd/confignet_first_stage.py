"""ConfigNetFirstStage (reference: confignet/confignet_first_stage.py): same config dict, attributes,
step functions, persistence format and inference API, with every network on HIP kernels.

Out of scope here: the controllability metrics and the TensorBoard / AzureML sinks (FID / KID and
the image-grid checkpoints are built: confignet_amd/metrics, run_checkpoints).  `train()` keeps the reference's iteration
structure and timing."""
import contextlib
import json
import os
import time
from collections import OrderedDict

import numpy as np
import torch

from . import confignet_utils, ops, optim, parallel
from .dnn_models.building_blocks import MLPSimple
from .dnn_models.hologan_discriminator import HologanDiscriminator, HologanLatentRegressor
from .dnn_models.hologan_generator import HologanGenerator
from .dnn_models.synthetic_encoder import SyntheticDataEncoder
from .losses import (GAN_G_loss, GAN_G_losses, compute_discriminator_loss, compute_latent_discriminator_loss,
                     compute_latent_regression_loss, discriminator_loss_fake, discriminator_loss_real, eye_loss)
from .neural_renderer_dataset import dump_pickle, load_pickle
from .losses import total as total_loss
from .nn import backward_into_arenas, require_gpu
from .perceptual_loss import PerceptualLoss

# the generator step deliberately runs one branch of the tape on a second stream (see _generator_loss)
if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

DEFAULT_CONFIG = {
    "model_type": None,
    "latent_dim": 128,
    "output_shape": (128, 128, 3),
    "const_input_shape": (4, 4, 4, 512),
    "n_adain_mlp_layers": 2,
    "n_adain_mlp_units": 128,
    "gen_output_activation": "tanh",
    "n_discr_features_at_layer_0": 48,
    "max_discr_filters": 512,
    "n_discr_layers": 5,
    "discr_conv_kernel_size": 3,
    "latent_regression_weight": 10.0,
    "use_style_discriminator": True,
    "rotation_ranges": ((-30, 30), (-10, 10), (0, 0)),
    "relu_before_in": True,
    "initial_from_rgb_layer_in_discr": True,
    "adain_on_learned_input": False,
    "latent_regressor_rot_weight": 5.0,
    "optimizer": {"lr": 0.0004, "beta_1": 0.0, "beta_2": 0.9, "amsgrad": False},
    "batch_size": 24,
    "n_discriminator_updates": 1,
    "n_generator_updates": 1,
    "latent_distribution": "normal",
    "metrics_checkpoint_period": 1000,
    "image_checkpoint_period": 500,
    "facemodel_inputs": {
        "texture_embedding": (None, 30),
        "geometry_identity_params": (None, 30),
        "blendshape_values": (None, 30),
        "beard_style_embedding": (None, 7),
        "eyebrow_style_embedding": (None, 7),
        "lower_eyelash_style": (None, 2),
        "upper_eyelash_style": (None, 2),
        "head_hair_style_embedding": (None, 9),
        "eye_color": (None, 3),
        "head_hair_color": (None, 3),
        "hdri_embedding": (None, 20),
        "bone_rotations:left_eye": (None, 2),
    },
    "num_synth_encoder_layers": 2,
    "n_latent_discr_layers": 4,
    "image_loss_weight": 0.00005,
    "eye_loss_weight": 5,
    "domain_adverserial_loss_weight": 5.0,
}


@contextlib.contextmanager
def frozen(*nets):
    """Networks whose weights are not in the step's trainable list: no filter gradients are computed."""
    for n in nets:
        n.requires_grad_(False)
    try:
        yield
    finally:
        for n in nets:
            n.requires_grad_(True)


class ConfigNetFirstStage:
    def __init__(self, config, initialize=True, seed=None):
        self.config = confignet_utils.merge_configs(DEFAULT_CONFIG, config)
        self.config["model_type"] = "ConfigNetFirstStage"
        self.device = require_gpu()
        self._rng = np.random.default_rng(seed)
        from .graphs import StaticBuffers
        self._bufs = StaticBuffers(self.device)
        self._graphs = {}
        self._deferred = None
        self._deferred_terms = []                    # loss terms differentiated by hand in _generator_update (global batch statistics)
        self._prestaged, self._stagers = {}, {}      # cross-iteration overlap of the discriminator steps (see overlap_discriminators)
        # second stage: real / synthetic branches of the generator step on two streams (not in deterministic mode: both
        # branches add into the generator's gradient slots, and the order of those adds would depend on the race)
        # (read at USE time -- `fork_generator_step` is a property: ops.set_deterministic() after construction must take effect)
        self._fork_generator_step = True
        self._work_streams = []
        self.use_graphs = False       # capture each step's device half into a HIP graph (single-GPU runs)

        self.generator = None
        self.generator_smoothed = None
        self.discriminator = None
        self.latent_regressor = None
        self.latent_discriminator = None
        self.synth_discriminator = None

        self.g_losses, self.d_losses, self.metrics = {}, {}, {}
        self.synth_d_losses, self.latent_d_losses = {}, {}

        # Remove the inputs that do not have a defined input dimension; sort by name (l.114-116)
        fm = {k: tuple(v) for k, v in self.config["facemodel_inputs"].items() if v[0] is not None}
        self.config["facemodel_inputs"] = OrderedDict(sorted(fm.items(), key=lambda t: t[0]))
        self.config["latent_dim"] = sum(v[1] for v in self.config["facemodel_inputs"].values())   # l.118-120

        self.synthetic_encoder = None
        self.facemodel_param_distributions = None
        self.perceptual_loss = PerceptualLoss(self.config["output_shape"], model_type="imagenet")

        if initialize:
            self.initialize_network()

    # ---- weights / persistence (confignet_first_stage.py:129-206) ---------------------------------
    def get_weights(self, return_tensors=False):
        return {
            "generator_weights": self.generator.get_weights(),
            "generator_smoothed_weights": self.generator_smoothed.get_weights(),
            "discriminator_weights": self.discriminator.get_weights(),
            "latent_regressor_weights": self.latent_regressor.get_weights(),
            "synthetic_encoder_weights": self.synthetic_encoder.get_weights(),
            "latent_discriminator_weights": self.latent_discriminator.get_weights(),
            "synth_discriminator_weights": self.synth_discriminator.get_weights(),
        }

    def set_weights(self, weights):
        self.generator.set_weights(weights["generator_weights"])
        self.generator_smoothed.set_weights(weights["generator_smoothed_weights"])
        self.discriminator.set_weights(weights["discriminator_weights"])
        self.latent_regressor.set_weights(weights["latent_regressor_weights"])
        self.synthetic_encoder.set_weights(weights["synthetic_encoder_weights"])
        self.latent_discriminator.set_weights(weights["latent_discriminator_weights"])
        self.synth_discriminator.set_weights(weights["synth_discriminator_weights"])

    def get_training_step_number(self):
        return 0 if "loss_sum" not in self.g_losses else len(self.g_losses["loss_sum"]) - 1

    def get_batch_size(self):
        return self.config["batch_size"]

    def get_log_dict(self):
        return {"g_losses": self.g_losses, "d_losses": self.d_losses, "metrics": self.metrics}

    def set_logs(self, log_dict):
        self.g_losses, self.d_losses, self.metrics = log_dict["g_losses"], log_dict["d_losses"], log_dict["metrics"]

    @staticmethod
    def _weights_to_npz(weights):
        out = {}
        for k, lst in weights.items():
            arr = np.empty(len(lst), dtype=object)
            arr[:] = lst
            out[k] = arr
        return out

    def save(self, output_dir, output_filename):
        """<name>.npz (one object array of the Keras-ordered weight list per network), <name>.json,
        <name>_facemodel_distr.pck -- the reference's layout (l.173-180)."""
        os.makedirs(output_dir, exist_ok=True)
        np.savez(os.path.join(output_dir, output_filename + ".npz"), **self._weights_to_npz(self.get_weights()))
        with open(os.path.join(output_dir, output_filename + ".json"), "w") as fp:
            json.dump(self.config, fp, indent=4)
        # pickled under the reference's class paths (confignet.neural_renderer_dataset.*), so either side loads the file
        dump_pickle(self.facemodel_param_distributions, os.path.join(output_dir, output_filename + "_facemodel_distr.pck"))

    @classmethod
    def load(cls, file_path):
        with open(file_path, "r") as fp:
            config = json.load(fp)
        model = cls(config)
        weights = np.load(os.path.splitext(file_path)[0] + ".npz", allow_pickle=True)
        model.set_weights(weights)
        log_file = os.path.splitext(file_path)[0] + "_log.json"
        if os.path.exists(log_file):
            with open(log_file, "r") as fp:
                model.set_logs(json.load(fp))
        distr = os.path.splitext(file_path)[0] + "_facemodel_distr.pck"
        model.facemodel_param_distributions = None
        if os.path.exists(distr):
            try:
                model.facemodel_param_distributions = load_pickle(distr)
            except Exception as e:        # e.g. an sklearn GaussianMixture pickled by an incompatible scikit-learn
                print("WARNING: facemodel param distributions could not be unpickled (%r)" % (e,))
        if model.facemodel_param_distributions is None:
            print("WARNING: facemodel param distributions not loaded")
        return model

    @property
    def facemodel_input_dim(self):
        return sum(d for d, _ in self.config["facemodel_inputs"].values())

    def get_facemodel_param_idxs_in_latent(self, param_name):
        dims = list(self.config["facemodel_inputs"].values())
        names = list(self.config["facemodel_inputs"].keys())
        i = names.index(param_name)
        start = int(np.sum([x[1] for x in dims[:i]]))
        return range(start, start + dims[i][1])

    def set_facemodel_param_in_latents(self, latents, param_name, param_value):
        param_value = np.array(param_value)
        if len(param_value.shape) == 1:
            param_value = param_value[np.newaxis]
        latents_for_param = self.synthetic_encoder.per_facemodel_input_mlps[param_name].predict(param_value)
        new_latents = np.copy(latents)
        new_latents[:, self.get_facemodel_param_idxs_in_latent(param_name)] = latents_for_param
        return new_latents

    def _get_generator_kwargs(self):
        return {
            "latent_dim": self.config["latent_dim"],
            "output_shape": tuple(self.config["output_shape"][:2]),
            "n_adain_mlp_units": self.config["n_adain_mlp_units"],
            "n_adain_mlp_layers": self.config["n_adain_mlp_layers"],
            "gen_output_activation": self.config["gen_output_activation"],
        }

    def initialize_network(self):
        """confignet_first_stage.py:251-287."""
        rng = self._rng
        self.synthetic_encoder = SyntheticDataEncoder(self.config["facemodel_inputs"],
                                                      self.config["num_synth_encoder_layers"], rng=rng)
        dargs = {
            "img_shape": tuple(self.config["output_shape"][:2]),
            "num_resample": self.config["n_discr_layers"],
            "disc_kernel_size": self.config["discr_conv_kernel_size"],
            "disc_expansion_factor": self.config["n_discr_features_at_layer_0"],
            "disc_max_feature_maps": self.config["max_discr_filters"],
            "initial_from_rgb_layer_in_discr": self.config["initial_from_rgb_layer_in_discr"],
        }
        self.discriminator = HologanDiscriminator(rng=rng, **dargs)
        self.synth_discriminator = HologanDiscriminator(rng=rng, **dargs)
        L = self.config["latent_dim"]
        self.latent_discriminator = MLPSimple(self.config["n_latent_discr_layers"], L, L, 1, rng=rng)
        self.latent_regressor = HologanLatentRegressor(L, rng=rng, **dargs)
        self.generator = HologanGenerator(rng=rng, **self._get_generator_kwargs())
        self.generator_smoothed = HologanGenerator(rng=rng, **self._get_generator_kwargs())
        self.generator_smoothed.copy_weights_from(self.generator)

    def all_networks(self):
        return [self.generator, self.generator_smoothed, self.discriminator, self.synth_discriminator,
                self.latent_discriminator, self.latent_regressor, self.synthetic_encoder]

    # ---- training code ----------------------------------------------------------------------------
    def update_smoothed_weights(self, smoother_alpha=0.999):
        """w_bar = a*w_bar + (1-a)*w over the whole generator arena in one launch (l.393-400; the
        reference round-trips every weight through numpy)."""
        ops.ema_step(self.generator_smoothed.arena, self.generator.arena, smoother_alpha)
        self.generator_smoothed.mark_updated()      # raw-pointer kernel: derived filter copies / inference graphs are keyed on the epoch

    def sample_rotations(self, n_samples, axes=[0, 1, 2]):
        r = np.zeros((n_samples, 3))
        for axis in axes:
            lo, hi = self.config["rotation_ranges"][axis]
            r[:, axis] = np.pi * np.random.uniform(lo, hi, n_samples) / 180
        return r.astype(np.float32)

    def sample_latent_vector(self, n_samples):
        if self.config["latent_distribution"] == "normal":
            return np.random.normal(0, 1, (n_samples, self.config["latent_dim"]))
        elif self.config["latent_distribution"] == "uniform":
            return np.random.uniform(-1, 1, (n_samples, self.config["latent_dim"]))

    def sample_facemodel_params(self, n_samples):
        return [self.facemodel_param_distributions[name].sample(n_samples)[0]
                for name in self.config["facemodel_inputs"].keys()]

    def sample_synthetic_dataset(self, dataset, n_samples):
        """Host (numpy) sampler with the reference's return contract (l.425-435)."""
        idx = np.random.randint(0, dataset.imgs.shape[0], n_samples)
        params = [dataset.metadata_inputs[name][idx] for name in self.config["facemodel_inputs"].keys()]
        rot = dataset.metadata_inputs["rotations"][idx].astype(np.float32)
        return params, rot, np.copy(dataset.imgs[idx]).astype(np.float32), np.copy(dataset.eye_masks[idx])

    # ---- step machinery: host half (sampling + upload into static buffers) / device half (graph-capturable) --
    # The uint8 image pools live in HBM; batches are gathered/normalised/flipped by one kernel (data-path row of
    # SURVEY.md 8f).  The numpy RNG calls mirror the reference's order.
    def _pool(self, dataset):
        cache = getattr(dataset, "_cn_device_pool", None)
        if cache is None or cache["device"] != self.device:
            cache = {
                "device": self.device,
                "imgs": torch.as_tensor(np.ascontiguousarray(dataset.imgs)).to(self.device),
                "eye_masks": (torch.as_tensor(np.ascontiguousarray(dataset.eye_masks)).to(self.device)
                              if getattr(dataset, "eye_masks", None) is not None else None),
            }
            dataset._cn_device_pool = cache
        return cache

    # Four streams for the whole model, chosen on first use: one carries the main line of an iteration (host staging
    # copies, the generator-step replay, EMA), the discriminator-type steps capture and replay on one each, and the
    # generator step reuses "d"'s for its capture and "sd"'s for its real-image branch (the two phases never overlap).
    # Streams share the few hardware queues (GPU_MAX_HW_QUEUES = 4) and two replay streams on one queue serialise the
    # concurrent phase (scripts/stream_probe.py: 31 -> 40 ms, depending on how many unrelated streams the process had
    # created before), so the four are picked by measurement (graphs.independent_streams).
    _WORK_SLOTS = {"main": 0, "d": 1, "sd": 2, "ld": 3, "g": 1}

    def _work_stream(self, name):
        if name == "g" and ops.DETERMINISTIC:
            # deterministic mode: the library's per-stream workspaces (partial sums) are bound at capture time to the stream a
            # graph is captured on; the generator step's graph replays NEXT TO the discriminator step whose stream it would
            # otherwise share for its capture, so it gets a capture stream (and with it a workspace) of its own
            if getattr(self, "_g_capture_stream", None) is None:
                self._g_capture_stream = torch.cuda.Stream()
                self._release_on_exit([self._g_capture_stream])
            return self._g_capture_stream
        if not self._work_streams:
            from .graphs import independent_streams
            self._work_streams = independent_streams(4)
            self._release_on_exit(self._work_streams)
        return self._work_streams[self._WORK_SLOTS.get(name, 1)]

    @contextlib.contextmanager
    def _main_line(self):
        """Runs the enclosed iteration on the model's own main-line stream (graph mode), ordered after the caller's
        stream on entry and before it on exit."""
        if not self.use_graphs:
            yield
            return
        caller, own = torch.cuda.current_stream(), self._work_stream("main")
        if caller == own:
            yield
            return
        own.wait_stream(caller)
        with torch.cuda.stream(own):
            yield
        caller.wait_stream(own)

    @property
    def _branch_stream(self):
        return self._work_stream("sd")

    def _dev(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def _stage(self, key, arr, dtype=torch.float32):
        return self._bufs.stage(key, arr, dtype)

    def _stage_real(self, key, dataset, n):
        """Host half of a real-image batch: indices + flip flags (flip_random_subset_of_images)."""
        self._stage(key + "/real_idx", np.random.randint(0, dataset.imgs.shape[0], n), torch.int64)
        self._stage(key + "/real_flip", np.random.randint(0, 2, size=n), torch.uint8)

    def _real_imgs(self, key, dataset):
        return ops.gather_images_u8(self._pool(dataset)["imgs"], self._bufs[key + "/real_idx"], self._bufs[key + "/real_flip"])

    def _stage_synth(self, key, dataset, n, late=None):
        """late: a stream for the copies of everything but the image indices (see _prelaunch_generator_targets: the index buffer
        is read by gathers only, the parameter / rotation buffers may still be read by the step that is running)."""
        idx = np.random.randint(0, dataset.imgs.shape[0], n)
        self._stage(key + "/synth_idx", idx, torch.int64)
        with (torch.cuda.stream(late) if late is not None else contextlib.nullcontext()):
            for name in self.config["facemodel_inputs"].keys():
                self._stage(key + "/p/" + name, dataset.metadata_inputs[name][idx])
            self._stage(key + "/rot", dataset.metadata_inputs["rotations"][idx])

    def _synth_batch(self, key, dataset, imgs=True):
        params = [self._bufs[key + "/p/" + name] for name in self.config["facemodel_inputs"].keys()]
        rot = self._bufs[key + "/rot"]
        if not imgs:
            return params, rot, None, None
        idx = self._bufs[key + "/synth_idx"]
        pool = self._pool(dataset)
        return params, rot, ops.gather_images_u8(pool["imgs"], idx, None), pool["eye_masks"][idx].contiguous()

    def _graph_key(self, name, datasets, optimizer):
        return (name, tuple(id(d) for d in datasets), id(optimizer), self._bufs.generation)

    def _graph_of(self, name, datasets, optimizer):
        return self._graphs.get(self._graph_key(name, datasets, optimizer))

    def _release_on_exit(self, streams):
        """When this model goes away (its captured graphs with it) the library may re-bind the deterministic-mode workspaces of
        its streams (cn_det_release_stream; the streams themselves are kept alive by the finalizer until then)."""
        import weakref

        def release(streams=list(streams)):
            from ._lib import lib
            for st in streams:
                lib.cn_det_release_stream(st.cuda_stream)
        weakref.finalize(self, release)

    @property
    def fork_generator_step(self):
        return self._fork_generator_step and not ops.DETERMINISTIC

    @fork_generator_step.setter
    def fork_generator_step(self, on):
        self._fork_generator_step = bool(on)

    def _run_step(self, name, datasets, optimizer, device_fn):
        """optimizer.advance() on the host, then the device half -- eagerly, or as a captured HIP graph."""
        optimizer.advance(name)
        fn = device_fn

        def device_fn():
            # loss scalars are returned detached: keeping the tape alive would keep the leaves' AccumulateGrad
            # nodes (and their stream binding) alive across steps
            ops.zero_pool_begin(name, self.device)
            self._deferred_terms = []      # a term left behind by an aborted step must not join this step's backward pass
            try:
                return {k: v.detach() for k, v in fn().items()}
            finally:
                ops.zero_pool_end()
        if not self.use_graphs:
            return device_fn()
        key = self._graph_key(name, datasets, optimizer)
        g = self._graphs.get(key)
        if g is None:
            from .graphs import StepGraph
            self._graphs = {k: v for k, v in self._graphs.items() if k[3] == self._bufs.generation}
            g = self._graphs[key] = StepGraph(device_fn, stream=self._work_stream(name))
            g.name = name
        if self._deferred is not None:
            if g.graph is not None:
                self._deferred.append(g)       # replayed together with its independent sibling steps
                return g.result()              # (filled by g.finish() right after the replay)
            self._flush_deferred()             # not captured yet: everything collected before it runs first, in order
        if getattr(g, "prelaunched", False):   # called on its own while its real half is already in flight: wait for it, run the rest
            g.prelaunched = False
            torch.cuda.current_stream().wait_stream(g.stream)
            res = g.result()
            g.replay(g.early_cut)
            g.finish()
            return res
        return g()

    def run_concurrently(self, step_calls, then=None):
        """Runs step functions that do not depend on each other (the three discriminator-type steps of one
        iteration: each updates only its own network and reads generator/encoder weights that stay fixed until
        the generator step) as HIP graphs replayed on separate streams, so their many small launches overlap on
        the 256 CUs.  Host halves (sampling, staging, step counters) still run in the reference's order.
        `then`: the step that follows them (the generator step).  Its graph is cut where it first needs a discriminator
        (graphs.segment_break(early=True)); the part before the cut -- generator, encoder and VGG forward passes, heavy
        MFMA-bound kernels -- is replayed NEXT TO the discriminator-type graphs (latency-bound small launches), the rest after
        them: same values as the reference's order, since the early part reads nothing the discriminator steps write."""
        if not self.use_graphs:
            outs = [c() for c in step_calls]
            return outs + [then()] if then is not None else outs
        self._deferred = []
        try:
            outs = [c() for c in step_calls]
            n_siblings = len(self._deferred)
            if then is not None:
                self._deferred_then = n_siblings == len(step_calls)      # (else: warm-up / capture iteration, strictly in order)
                outs.append(then())
            self._flush_deferred()
        finally:
            self._deferred, self._deferred_then = None, False
        return outs

    _deferred_then = False

    def _flush_deferred(self):
        """Launch the graphs collected so far: each on the stream it was captured on (one of the model's measured-independent
        work streams, see _work_stream), a trailing generator-step graph's early segments on the calling stream beside them."""
        pending, self._deferred = self._deferred, []
        if not pending:
            return
        follower = pending.pop() if (self._deferred_then and len(pending) > 1 and pending[-1].early_cut) else None
        cur = torch.cuda.current_stream()
        for g in pending:
            g.stream.wait_stream(cur)
            with torch.cuda.stream(g.stream):
                if getattr(g, "prelaunched", False):       # its real half ran next to the previous iteration's generator tail
                    g.replay(g.early_cut)
                    g.prelaunched = False
                else:
                    g.replay()
                g.finish()
        ev_early = None
        if follower is not None and self.early_generator_forward:
            follower.replay(0, follower.early_cut)
            ev_early = torch.cuda.Event()
            ev_early.record(cur)
        for g in pending:
            cur.wait_stream(g.stream)
        if follower is not None:
            follower.replay(follower.early_cut if self.early_generator_forward else 0)
            follower.finish()
            if self.overlap_discriminators:
                # next iteration's image-discriminator steps: host half (np.random draws in the reference's order: after the
                # generator step's, discriminator before synthetic discriminator; staging copies on the step's own stream,
                # behind the work it has just finished) and the real half of the graph -- they run under the generator tail.
                # (AFTER the tail has been issued: the tail's forked branch shares the synthetic discriminator's stream, and a
                # real half queued in front of it holds the whole tail back -- starting the halves right after each step's own
                # update was tried: 324 -> 291 images/s)
                for g in pending:
                    st = self._stagers.get(getattr(g, "name", None))
                    if st is None or not g.early_cut:
                        continue
                    stage, training_set, optimizer = st
                    with torch.cuda.stream(g.stream):
                        stage(training_set)
                        g.replay(0, g.early_cut)
                    g.prelaunched = True
                    net = self.discriminator if g.name == "d" else self.synth_discriminator
                    self._prestaged[g.name] = (id(training_set), id(optimizer), net.epoch, self._bufs.generation)
                self._prelaunch_generator_targets(pending, follower, ev_early)

    def _prelaunch_generator_targets(self, pending, follower, ev_early):
        """(Second stage; no-op otherwise.)"""

    early_generator_forward = True

    # Cross-iteration overlap (graph dispatch, steady training loops: bench.py and train() switch it on).  The REAL half of
    # an image-discriminator step -- forward on real images, the R1 sweep and tangent pass, their backward: more than half of
    # the step -- reads only that discriminator's own weights, which are final when the discriminator phase of the previous
    # iteration ends.  With this flag the step's graph is cut between its halves, and the real half of iteration t+1 is
    # replayed on the step's stream NEXT TO the generator tail of iteration t (the host half of the step -- its np.random
    # draws and the staging copies -- moves with it: the draws still happen in the reference's order, generator step of t,
    # then discriminator steps of t+1).  The fake half and the Adam update follow the generator update as before.  Same
    # arithmetic; the two halves' gradients are added in the arena instead of inside one backward pass.
    overlap_discriminators = False

    def _discriminator_update(self, net, real_imgs, fake_imgs, optimizer, slot="default"):
        """real_imgs / fake_imgs: tensors, or (with overlap_discriminators) callables that produce them."""
        if callable(real_imgs):
            from .graphs import segment_break
            real, gp = discriminator_loss_real(net, real_imgs())
            backward_into_arenas(total_loss(list(real.values()) + list(gp.values())), [net])
            segment_break(early=True)          # nothing above reads a generator / encoder weight
            fake = discriminator_loss_fake(net, fake_imgs())
            backward_into_arenas(total_loss(fake.values()), [net], accumulate=True)
            optimizer.apply_gradients(net, advance=False, slot=slot)
            losses = {**real, **fake, **gp}    # the reference's key order (losses.py:20-47)
            losses["loss_sum"] = total_loss(losses.values())
            return losses
        losses = compute_discriminator_loss(net, real_imgs, fake_imgs)
        backward_into_arenas(losses["loss_sum"], [net])
        optimizer.apply_gradients(net, advance=False, slot=slot)
        return losses

    def get_discriminator_batch(self, training_set):
        """Reference-shaped helper (confignet_first_stage.py:438-450): returns (real_imgs, fake_imgs)."""
        self._stage_d_batch(training_set)
        return self._d_batch(training_set)

    def _stage_d_batch(self, training_set):
        n = self.get_batch_size()
        self._stage_real("d", training_set, n)
        self._stage("d/z", self.sample_latent_vector(n))
        self._stage("d/rot", self.sample_rotations(n))

    def _d_fake(self, training_set):
        with torch.no_grad():
            return self.generator([self._bufs["d/z"], self._bufs["d/rot"]])

    def _d_batch(self, training_set):
        return self._real_imgs("d", training_set), self._d_fake(training_set)

    def _image_discriminator_step(self, name, net, training_set, optimizer, stage, real, fake):
        """Host half + device half of the discriminator / synthetic-discriminator step (l.466-516)."""
        pre = self._prestaged.pop(name, None)
        if pre != (id(training_set), id(optimizer), net.epoch, self._bufs.generation):
            # no real half in flight for exactly this call (first iteration, other arguments, weights replaced since): the
            # step starts from its host half; a half that was pre-replayed for something else is simply redone
            for g in self._graphs.values():
                if getattr(g, "name", None) == name:
                    g.prelaunched = False
            stage(training_set)
        self._stagers[name] = (stage, training_set, optimizer)
        if self.overlap_discriminators and self.use_graphs:
            device = lambda: self._discriminator_update(net, lambda: real(training_set), lambda: fake(training_set), optimizer, name)
        else:
            device = lambda: self._discriminator_update(net, real(training_set), fake(training_set), optimizer, name)
        return self._run_step(name, (training_set,), optimizer, device)

    def discriminator_training_step(self, training_set, optimizer):
        return self._image_discriminator_step("d", self.discriminator, training_set, optimizer, self._stage_d_batch,
                                              lambda ts: self._real_imgs("d", ts), self._d_fake)

    def _stage_sd_batch(self, training_set):
        n = self.get_batch_size()
        self._stage_real("sd", training_set, n)
        self._stage_synth("sd", training_set, n)

    def _sd_fake(self, training_set):
        params, rotations, _, _ = self._synth_batch("sd", training_set, imgs=False)
        with torch.no_grad():
            return self.generator([self.synthetic_encoder(params), rotations])

    def _sd_batch(self, training_set):
        return self._real_imgs("sd", training_set), self._sd_fake(training_set)

    def synth_discriminator_training_step(self, synth_training_set, optimizer):
        return self._image_discriminator_step("sd", self.synth_discriminator, synth_training_set, optimizer, self._stage_sd_batch,
                                              lambda ts: self._real_imgs("sd", ts), self._sd_fake)

    def _latent_discriminator_update(self, real_latents, fake_latents, optimizer):
        net = self.latent_discriminator
        losses = compute_latent_discriminator_loss(net, real_latents, fake_latents)
        backward_into_arenas(losses["loss_sum"], [net])
        optimizer.apply_gradients(net, advance=False, slot="ld")
        return losses

    def latent_discriminator_training_step(self, synth_training_set, optimizer):
        n = self.get_batch_size()
        self._stage("ld/z", self.sample_latent_vector(n))
        self._stage_synth("ld", synth_training_set, n)

        def device():
            params, _, _, _ = self._synth_batch("ld", synth_training_set, imgs=False)
            with torch.no_grad():
                fake_latents = self.synthetic_encoder(params)
            return self._latent_discriminator_update(self._bufs["ld/z"], fake_latents, optimizer)
        return self._run_step("ld", (synth_training_set,), optimizer, device)

    def _generator_loss(self, facemodel_params, synth_rotations, gt_imgs, eye_masks, real_latents, real_rotations):
        """The taped part of generator_training_step (l.518-554)."""
        cfg = self.config
        losses = {}
        # the sampled-latent branch (generator -> discriminator) runs on a second stream next to the synthetic branch
        # (see ConfigNet._generator_loss); both meet again in the latent regressor
        main = torch.cuda.current_stream()
        side = self._branch_stream if self.fork_generator_step else main
        if side is not main:
            side.wait_stream(main)
        from .graphs import segment_break
        with torch.cuda.stream(side):
            generator_output_real = self.generator((real_latents, real_rotations))
        synth_latents = self.synthetic_encoder(facemodel_params)
        generator_output_synth = self.generator((synth_latents, synth_rotations))
        losses["image_loss"] = cfg["image_loss_weight"] * self.perceptual_loss.loss(gt_imgs, generator_output_synth)
        losses["eye_loss"] = cfg["eye_loss_weight"] * eye_loss(gt_imgs, generator_output_synth, eye_masks)
        if side is not main:
            main.wait_stream(side)
        segment_break(early=True)              # nothing above reads a discriminator weight
        if side is not main:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            gan_real = GAN_G_losses(self.discriminator(generator_output_real).values())
        for i, l in enumerate(GAN_G_losses(self.synth_discriminator(generator_output_synth).values())):
            losses["GAN_loss_synth_" + str(i)] = l
        if side is not main:
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():    # (see ConfigNet._generator_loss: allocator hand-off in eager mode)
                for t in [generator_output_real] + gan_real:
                    t.record_stream(main)
        for i, l in enumerate(gan_real):
            losses["GAN_loss_real_" + str(i)] = l
        latent_discriminator_output = self.latent_discriminator(synth_latents)
        losses["latent_GAN_loss"] = cfg["domain_adverserial_loss_weight"] * GAN_G_loss(latent_discriminator_output)
        stacked_latents = torch.cat((synth_latents, real_latents), dim=0)
        stacked_imgs = torch.cat((generator_output_synth, generator_output_real), dim=0)
        stacked_rotations = torch.cat((synth_rotations, real_rotations), dim=0)
        labels = torch.cat((stacked_latents, cfg["latent_regressor_rot_weight"] * stacked_rotations), dim=-1)
        reg = compute_latent_regression_loss(stacked_imgs, labels, self.latent_regressor)
        losses["latent_regression_loss"] = cfg["latent_regression_weight"] * reg
        losses["loss_sum"] = total_loss(losses.values())
        return losses

    def _generator_update(self, losses, nets, optimizer, cut=None):
        """tape.gradient + apply_gradients of the generator step.  Data parallel with `cut` = (tensors, late_nets): the
        networks in late_nets (the real encoder) are reached only through `tensors` (its outputs), so the backward pass
        is taken in two parts -- everything down to `tensors` first, whose gradient arenas (generator, latent regressor,
        synthetic encoder: 62 MB) then start their RCCL all-reduce while the second part (the ResNet-50 backward, the last
        and longest stretch of the tape) is still computing; only the encoder's arena is exchanged after it."""
        roots, cots = losses["loss_sum"], None
        terms, self._deferred_terms = self._deferred_terms, []
        if terms:
            # loss terms with global batch statistics: their (tensor, cotangent) pairs join the scalar as roots of the backward pass
            pairs = [pc for t in terms for pc in t.cotangents()]
            roots = [losses["loss_sum"]] + [t for t, _ in pairs]
            cots = [torch.ones_like(losses["loss_sum"])] + [c for _, c in pairs]
        if cut is None or not parallel.active():
            backward_into_arenas(roots, nets, grad_outputs=cots)
        else:
            from .graphs import segment_break
            tensors, late = cut
            early = [n for n in nets if all(n is not m for m in late)]
            cut_grads = backward_into_arenas(roots, early, extra=tensors, grad_outputs=cots)
            segment_break(lambda: parallel.begin_allreduce(early))
            live = [(t, g) for t, g in zip(tensors, cut_grads) if g is not None]
            backward_into_arenas([t for t, _ in live], late, grad_outputs=[g for _, g in live])
        optimizer.apply_gradients(nets, advance=False, slot="g")

    def generator_training_step(self, real_training_set, synth_training_set, optimizer):
        n_synth = self.get_batch_size() // 2
        n_real = self.get_batch_size() - n_synth
        assert n_synth > 0, "the generator step splits the batch into a synthetic and a real half: batch_size >= 2 " \
                            "(the reference's losses are means over an empty synthetic batch, i.e. NaN, at batch_size 1)"
        self._stage_synth("g", synth_training_set, n_synth)
        self._stage("g/z", self.sample_latent_vector(n_real))
        self._stage("g/rot_real", self.sample_rotations(n_real))
        nets = [self.generator, self.latent_regressor, self.synthetic_encoder]

        def device():
            params, synth_rot, gt_imgs, eye_masks = self._synth_batch("g", synth_training_set)
            with frozen(self.discriminator, self.synth_discriminator, self.latent_discriminator):
                losses = self._generator_loss(params, synth_rot, gt_imgs, eye_masks, self._bufs["g/z"], self._bufs["g/rot_real"])
                self._generator_update(losses, nets, optimizer)
            return losses
        return self._run_step("g", (real_training_set, synth_training_set), optimizer, device)

    n_checkpoint_rotations = 6       # confignet_first_stage.py:111-112
    n_checkpoint_samples = 10

    def setup_training(self, log_dir, synth_training_set, n_samples_for_metrics, real_training_set=None):
        """confignet_first_stage.py:562-595, same order of np.random draws: the FID/KID reference sample, the metric and
        checkpoint-visualisation generator inputs, the synthetic checkpoint batch.  config["run_metrics"] = False skips the
        InceptionV3 extractor (and only that)."""
        if real_training_set is None:
            real_training_set = synth_training_set
        if log_dir is not None:
            os.makedirs(log_dir, exist_ok=True)
        self._inception_metric_object = None
        if self.config.get("run_metrics", True):
            from .metrics import InceptionMetrics
            self._inception_metric_object = InceptionMetrics(self.config, real_training_set)
        self._generator_input_for_metrics = {"latent": self.sample_latent_vector(n_samples_for_metrics),
                                             "rotation": self.sample_rotations(n_samples_for_metrics)}
        checkpoint_latent = np.vstack([self.sample_latent_vector(self.n_checkpoint_samples)] * self.n_checkpoint_rotations)
        checkpoint_rotation = np.zeros((self.n_checkpoint_rotations, 3))
        rr = self.config["rotation_ranges"][0]
        checkpoint_rotation[:, 0] = np.pi * np.linspace(rr[0], rr[1], self.n_checkpoint_rotations) / 180
        checkpoint_rotation = np.reshape(np.hstack([checkpoint_rotation] * self.n_checkpoint_samples), (-1, 3))
        self._checkpoint_visualization_input = {"latent": checkpoint_latent, "rotation": checkpoint_rotation}
        self.facemodel_param_distributions = synth_training_set.metadata_input_distributions   # l.587
        facemodel_params, _, gt_imgs, _ = self.sample_synthetic_dataset(synth_training_set, self.n_checkpoint_samples)
        facemodel_params = [np.tile(np.asarray(p), (self.n_checkpoint_rotations, 1)) for p in facemodel_params]
        self._checkpoint_visualization_input["facemodel_params"] = facemodel_params
        self._checkpoint_visualization_input["gt_imgs"] = gt_imgs

    # ---- checkpoint-related code (confignet_first_stage.py:289-386) ----------------------------------
    def synth_data_image_checkpoint(self, output_dir):
        vis = self._checkpoint_visualization_input
        generated = self.generate_images_from_facemodel(vis["facemodel_params"], vis["rotation"])
        gt = vis["gt_imgs"]
        if torch.is_tensor(gt):
            gt = gt.detach().cpu().numpy()
        gt = np.asarray(gt)
        if gt.dtype != np.uint8:                               # the device pipeline hands out [-1, 1] floats
            gt = np.clip((gt + 1.0) * 127.5, 0, 255).astype(np.uint8)
        grid = confignet_utils.build_image_matrix(np.vstack((gt, generated)), self.n_checkpoint_rotations + 1, self.n_checkpoint_samples)
        img_dir = os.path.join(output_dir, "output_imgs")
        os.makedirs(img_dir, exist_ok=True)
        confignet_utils.write_image(os.path.join(img_dir, str(self.get_training_step_number()).zfill(6) + "_synth.jpg"), grid)

    def image_checkpoint(self, output_dir):
        vis = self._checkpoint_visualization_input
        grid = confignet_utils.build_image_matrix(self.generate_images(vis["latent"], vis["rotation"]),
                                                  self.n_checkpoint_rotations, self.n_checkpoint_samples)
        img_dir = os.path.join(output_dir, "output_imgs")
        os.makedirs(img_dir, exist_ok=True)
        confignet_utils.write_image(os.path.join(img_dir, str(self.get_training_step_number()).zfill(6) + ".png"), grid)
        self.synth_data_image_checkpoint(output_dir)

    def generate_output_for_metrics(self):
        return self.generate_images(self._generator_input_for_metrics["latent"], self._generator_input_for_metrics["rotation"])

    def calculate_metrics(self, output_dir, aml_run=None):
        """l.378-386: KID / FID of n_samples_for_metrics generated images against the training-set sample."""
        if self._inception_metric_object is None:
            return
        generated_images = self.generate_output_for_metrics()
        self.metrics.setdefault("training_step_number", []).append(self.get_training_step_number())
        self._inception_metric_object.update_and_log_metrics(generated_images, self.metrics, output_dir, aml_run, None)

    def run_checkpoints(self, output_dir, iteration_time, aml_run=None, checkpoint_start=None):
        """l.334-376: every image_checkpoint_period steps loss logs + image grids, every metrics_checkpoint_period steps
        metrics + save (step 0 included).  Rank 0 only under data parallelism."""
        if output_dir is None or parallel.rank() != 0:
            return
        checkpoint_start = time.perf_counter()
        step_number = self.get_training_step_number()
        if step_number % self.config["image_checkpoint_period"] == 0:
            confignet_utils.log_loss_vals(self.synth_d_losses, output_dir, step_number, "synth_discriminator_", aml_run=aml_run)
            confignet_utils.log_loss_vals(self.latent_d_losses, output_dir, step_number, "latent_discriminator_", aml_run=aml_run)
        if step_number % self.config["metrics_checkpoint_period"] == 0:
            print("Running metrics")
            self.calculate_metrics(output_dir, aml_run=aml_run)
            self.save(os.path.join(output_dir, "checkpoints"), str(step_number).zfill(6))
        if step_number % self.config["image_checkpoint_period"] == 0:
            self.image_checkpoint(output_dir)
            confignet_utils.log_loss_vals(self.g_losses, output_dir, step_number, "generator_", aml_run=aml_run)
            confignet_utils.log_loss_vals(self.d_losses, output_dir, step_number, "discriminator_", aml_run=aml_run)
            print("Training iteration time: %f" % iteration_time)
            print("Checkpoint time: %f" % (time.perf_counter() - checkpoint_start))

    def train(self, real_training_set, synth_training_set, output_dir, log_dir, n_steps=100000,
              n_samples_for_metrics=1000, aml_run=None):
        """confignet_first_stage.py:597-626."""
        self.setup_training(log_dir, synth_training_set, n_samples_for_metrics, real_training_set=real_training_set)
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
            with self._main_line():
                d_steps = [lambda: self.discriminator_training_step(real_training_set, discriminator_optimizer),
                           lambda: self.synth_discriminator_training_step(synth_training_set, discriminator_optimizer),
                           lambda: self.latent_discriminator_training_step(synth_training_set, discriminator_optimizer)]
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
            torch.cuda.synchronize()
            self.last_iteration_time = time.perf_counter() - t0
            print("[D loss: %f] [synth_D loss: %f] [latent_D_loss: %f] [G loss: %f]" %
                  (d_loss["loss_sum"], synth_d_loss["loss_sum"], latent_d_loss["loss_sum"], g_loss["loss_sum"]))
            confignet_utils.update_loss_dict(self.g_losses, g_loss)
            confignet_utils.update_loss_dict(self.d_losses, d_loss)
            confignet_utils.update_loss_dict(self.synth_d_losses, synth_d_loss)
            confignet_utils.update_loss_dict(self.latent_d_losses, latent_d_loss)
            self.run_checkpoints(output_dir, self.last_iteration_time, aml_run=aml_run)

    # ---- evaluation code ----------------------------------------------------------------------------
    use_inference_graphs = True      # generate_images: one replayed HIP graph per (generator, batch) instead of ~60 eager launches

    def _generate_images_with(self, generator, latent_vector, rotations):
        inp = self.generator.build_input_dict(latent_vector, rotations)
        n = len(inp["rotation"])
        host = {}
        for k, v in inp.items():                                            # one float32 host array per DISTINCT input object
            if id(v) not in host:
                host[id(v)] = v if torch.is_tensor(v) else np.ascontiguousarray(v, dtype=np.float32)
        inp = {k: host[id(v)] for k, v in inp.items()}
        outs = []
        with torch.no_grad():
            for s in range(0, n, 32):                                       # keras predict batch 32 (R11)
                chunk = {k: (v if n <= 32 else v[s:s + 32]) for k, v in inp.items()}
                if self.use_inference_graphs:
                    img = self._replay_generator(generator, chunk)
                else:
                    img = ops.to_uint8(generator(chunk))                    # clip + (x+1)*127.5 -> uint8
                outs.append(img.cpu().numpy())
        return outs[0] if len(outs) == 1 else np.concatenate(outs, axis=0)

    def _replay_generator(self, generator, chunk):
        """uint8 images of one <= 32 chunk through a cached InferenceGraph (generator forward + to_uint8)."""
        from .graphs import InferenceGraph
        cache = self.__dict__.setdefault("_infer_graphs", {})
        key = (id(generator), len(chunk["rotation"]), generator.epoch, ops.ACT_DTYPE)
        g = cache.get(key)
        if g is None:
            for k in [k for k in cache if k[0] == key[0] and k[2] != key[2]]:      # stale weights epoch
                del cache[k]
            if len(cache) >= 8:
                cache.pop(next(iter(cache)))
            dev_chunk = {k: generator.to_device(v) for k, v in chunk.items()}
            g = cache[key] = InferenceGraph(lambda **kw: ops.to_uint8(generator(kw)), dev_chunk, self._work_stream("main"))
        return g(**chunk)

    def generate_images(self, latent_vector, rotations):
        """confignet_first_stage.py:633-639."""
        return self._generate_images_with(self.generator_smoothed, latent_vector, rotations)

    def generate_images_from_facemodel(self, facemodel_params, rotations):
        with torch.no_grad():
            latents = self.synthetic_encoder(facemodel_params).cpu().numpy()
        return self.generate_images(latents, rotations)

    def fit_facemodel_expression_params_to_latent(self, latent, unused_expr_idxs=None, param_name="blendshape_values",
                                                  n_iters=2000, learning_rate=0.05, verbose=False):
        """The face-model parameter vector in [0, 1] whose per-input MLP of the synthetic encoder lands closest (mean squared
        error) to the `param_name` slice of `latent` (reference confignet_first_stage.py:646-679: plain SGD on one (1, d) variable,
        clipped to [0, 1] after every step, unused expressions zeroed).  The MLP runs on the HIP kernels; the whole loop stays on
        the device (the reference round-trips the variable through numpy every step)."""
        idxs = list(self.get_facemodel_param_idxs_in_latent(param_name))
        names = list(self.config["facemodel_inputs"].keys())
        n_in = list(self.config["facemodel_inputs"].values())[names.index(param_name)][0]
        mlp = self.synthetic_encoder.per_facemodel_input_mlps[param_name]
        target = self.synthetic_encoder.to_device(np.asarray(latent, dtype=np.float32)[:, idxs])
        values = torch.zeros((1, n_in), device=target.device, dtype=torch.float32, requires_grad=True)
        unused = None if unused_expr_idxs is None else torch.as_tensor(list(unused_expr_idxs), device=target.device, dtype=torch.long)
        for step in range(n_iters):
            loss = torch.mean(torch.square(target - mlp(values)))
            (grad,) = torch.autograd.grad(loss, [values])
            with torch.no_grad():
                values -= learning_rate * grad
                values.clamp_(0.0, 1.0)
                if unused is not None:
                    values[:, unused] = 0.0
            if verbose:
                print("%d: %f" % (step, float(loss)))
        return values.detach().cpu().numpy()
