#!/usr/bin/env python
"""Two-stage ConfigNet training on MI355X with the reference's command line (reference: train_confignet.py:15-73).

    python train_confignet.py --output_dir OUT --real_training_set_path real.pck --synth_training_set_path synth.pck \
        --validation_set_path val.pck --attribute_classifier_path none [--batch_size 24] [--stage_1_training_steps N]
    python -m torch.distributed.run --nproc-per-node 8 train_confignet.py ...      # data parallel, one rank per GPU

Dataset files are the reference's (`NeuralRendererDataset.save`: <name>.pck + <name>_imgs.dat).  `--synthetic N` replaces
all three by seeded FFHQ-shaped noise sets of N images (no dataset can be downloaded here).  Metrics that need the
attribute classifier / InceptionV3 are out of scope, so --attribute_classifier_path is accepted and ignored."""
import argparse
import os
import sys

import training_utils
import confignet
from confignet.confignet_first_stage import DEFAULT_CONFIG

FLAGS = [
    ("--output_dir", dict(required=True)), ("--log_dir", dict(default=None)), ("--data_dir", dict(default=None)),
    ("--real_training_set_path", dict(default=None)), ("--synth_training_set_path", dict(default=None)),
    ("--validation_set_path", dict(default=None)), ("--attribute_classifier_path", dict(default=None)),
    ("--batch_size", dict(type=int, default=DEFAULT_CONFIG["batch_size"])),
    ("--stage_1_training_steps", dict(type=int, default=50000)), ("--stage_2_training_steps", dict(type=int, default=100000)),
    ("--n_samples_for_metrics", dict(type=int, default=1000)),
    ("--synthetic", dict(type=int, default=0, help="use seeded synthetic datasets of this many images")),
    ("--resolution", dict(type=int, default=256, help="image size of the synthetic datasets")),
]


def parse_args(argv):
    ap = argparse.ArgumentParser(description="ConfigNet training (MI355X)")
    for flag, kw in FLAGS:
        ap.add_argument(flag, **kw)
    args = ap.parse_args(argv)
    from confignet_amd import parallel
    parallel.init_from_env()
    training_utils.initialize_random_seed(parallel.rank())            # per-rank batch sampling stream (rank 0: seed 0 as l.34)
    if args.synthetic:
        from confignet_amd import SyntheticFaceDataset
        real, synth, val = (SyntheticFaceDataset(args.synthetic, args.resolution, seed=s) for s in (1, 2, 3))
    else:
        paths = [args.real_training_set_path, args.synth_training_set_path, args.validation_set_path]
        assert all(paths), "dataset paths (or --synthetic N) are required"
        if args.data_dir is not None:
            paths = [os.path.join(args.data_dir, p) for p in paths]
        real, synth, val = (confignet.NeuralRendererDataset.load(p) for p in paths)
    log_dir = args.log_dir or args.output_dir
    config = confignet.confignet_utils.merge_configs(DEFAULT_CONFIG, {"batch_size": args.batch_size,
                                                                       "output_shape": tuple(real.imgs.shape[1:])})
    synth.process_metadata(config, True)

    first = confignet.ConfigNetFirstStage(config, seed=0)
    first.train(real, synth, os.path.join(args.output_dir, "first_stage"), log_dir, n_steps=args.stage_1_training_steps,
                n_samples_for_metrics=args.n_samples_for_metrics)
    weights = first.get_weights()

    config["image_loss_weight"] *= 10                                  # l.67
    second = confignet.ConfigNet(config, seed=0)
    confignet.ConfigNetFirstStage.set_weights(second, weights)         # l.69: called unbound on the second-stage model
    # (the reference passes stage_1_training_steps here too, l.72; --stage_2_training_steps is parsed and unused)
    second.train(real, synth, val, args.attribute_classifier_path, args.output_dir, log_dir,
                 n_steps=args.stage_1_training_steps, n_samples_for_metrics=args.n_samples_for_metrics)
    return second


if __name__ == "__main__":
    parse_args(sys.argv[1:])
