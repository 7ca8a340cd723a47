#!/usr/bin/env python
"""LatentGAN training on MI355X with the reference's command line (reference: train_latent_gan.py:10-48)."""
import argparse
import os
import sys

import training_utils
import confignet
from confignet.latent_gan import DEFAULT_CONFIG


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--confignet_path", required=True)
    ap.add_argument("--training_set_path", required=True)
    ap.add_argument("--output_dir", required=True)
    for key, typ in (("num_mlp_layers", int), ("hidden_layer_size_multiplier", float), ("latent_distribution_type", str),
                     ("batch_size", int)):
        ap.add_argument("--" + key, type=typ, default=DEFAULT_CONFIG[key])
    ap.add_argument("--n_training_steps", type=int, default=100000)
    ap.add_argument("--n_samples_for_metrics", type=int, default=1000)
    ap.add_argument("--data_dir", default=None)
    ap.add_argument("--log_dir", default=None)
    args = ap.parse_args(argv)
    training_utils.initialize_random_seed(0)
    join = (lambda p: os.path.join(args.data_dir, p)) if args.data_dir is not None else (lambda p: p)
    training_set = confignet.NeuralRendererDataset.load(join(args.training_set_path))
    model = confignet.load_confignet(join(args.confignet_path))
    config = {k: getattr(args, k) for k in ("num_mlp_layers", "latent_distribution_type", "hidden_layer_size_multiplier",
                                            "batch_size", "n_samples_for_metrics")}
    config["latent_dim"] = model.config["latent_dim"]
    gan = confignet.LatentGAN(config)
    gan.train(training_set, model, args.output_dir, args.log_dir or args.output_dir, n_iters=args.n_training_steps)
    return gan


if __name__ == "__main__":
    parse_args(sys.argv[1:])
