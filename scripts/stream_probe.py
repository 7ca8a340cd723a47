"""Robustness probe of the concurrent discriminator phase: create `k` extra streams (each used once) before / after the
model exists and report the phase times.  usage: stream_probe.py <k_before> <k_after>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from confignet_amd import ConfigNet, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs

kb, ka = int(sys.argv[1]), int(sys.argv[2])
keep = []


def extra(k):
    for _ in range(k):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            keep.append((st, torch.zeros(16, device="cuda") + 1))


torch.cuda.set_device(0)
extra(kb)
ds = SyntheticFaceDataset(64, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 16, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
m.setup_training(None, ds, 0, real_training_set=ds)
d, g = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
m.use_graphs = True
for _ in range(3):
    m.training_iteration(ds, ds, d, g)
extra(ka)
torch.cuda.synchronize()


def timed(fn, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        with m._main_line():
            fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


dphase = lambda: m.run_concurrently([lambda: m.discriminator_training_step(ds, d), lambda: m.synth_discriminator_training_step(ds, d),
                                     lambda: m.latent_discriminator_training_step(ds, ds, d)])
print("extra streams before=%d after=%d: D-phase %.2f ms, G %.2f ms, iteration %.2f ms" % (
    kb, ka, timed(dphase), timed(lambda: m.generator_training_step(ds, ds, g)), timed(lambda: m.training_iteration(ds, ds, d, g))), flush=True)
