#!/usr/bin/env python
"""Headless ConfigNet demo on MI355X (reference: evaluation/confignet_demo.py; no window -- frames go to --output_dir as
.npy canvases, keys come from --keys).  Models are loaded with the reference's layout (model.json + model.npz).

    python evaluation/confignet_demo.py --confignet_model_path models/confignet_256/model.json \
        --latent_gan_model_path models/latentgan_256/model.json --keys "ddwwx  b" [--image_path face.npy] [--test_mode]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from confignet import ConfigNet, LatentGAN   # noqa: E402
from confignet_amd.demo import DemoSession   # noqa: E402


def run(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--image_path", default=None, help=".npy with one (R,R,3) or several (M,R,R,3) aligned uint8 face images")
    ap.add_argument("--confignet_model_path", required=True)
    ap.add_argument("--latent_gan_model_path", default=None)
    ap.add_argument("--n_rows", type=int, default=2)
    ap.add_argument("--n_cols", type=int, default=3)
    ap.add_argument("--keys", default="", help="key presses, one per frame (space, wsad, ikjl, n, x, z, c, v, b)")
    ap.add_argument("--output_dir", default=None)
    ap.add_argument("--hdri_turntable_path", default=None, help="assets/hdri_turntable_embeddings.npy of the reference")
    ap.add_argument("--test_mode", action="store_true", help="one frame, every key handler fired once")
    args = ap.parse_args(argv)
    images = None
    if args.image_path is not None:
        arr = np.load(args.image_path)
        images = [arr] if arr.ndim == 3 else list(arr)
    latentgan = LatentGAN.load(args.latent_gan_model_path) if images is None else None
    model = ConfigNet.load(args.confignet_model_path)
    hdri = np.load(args.hdri_turntable_path) if args.hdri_turntable_path else None
    session = DemoSession(model, latentgan, images, args.n_rows, args.n_cols, hdri)
    count = [0]

    def save(canvas):
        if args.output_dir is not None:
            os.makedirs(args.output_dir, exist_ok=True)
            np.save(os.path.join(args.output_dir, "frame_%04d.npy" % count[0]), canvas)
        count[0] += 1
    session.run(test_mode=args.test_mode, keys=args.keys, on_frame=save)
    return session


if __name__ == "__main__":
    run(sys.argv[1:])
