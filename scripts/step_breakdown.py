"""Per-step-function GPU time of one second-stage iteration at 256x256, batch 16."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from confignet_amd import ConfigNet, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs

np.random.seed(0)
ds = SyntheticFaceDataset(64, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 16, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
m.setup_training(None, ds, 0, real_training_set=ds)
dopt, gopt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
steps = [("D", lambda: m.discriminator_training_step(ds, dopt)),
         ("synthD", lambda: m.synth_discriminator_training_step(ds, dopt)),
         ("latentD", lambda: m.latent_discriminator_training_step(ds, ds, dopt)),
         ("G", lambda: m.generator_training_step(ds, ds, gopt)),
         ("EMA", lambda: m.update_smoothed_weights())]
for _ in range(2):
    for _, f in steps:
        f()
torch.cuda.synchronize()
tot = {}
import time
for rep in range(3):
    for name, f in steps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        f()
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        tot.setdefault(name, []).append((e0.elapsed_time(e1), (t1 - t0) * 1e3))
for name, v in tot.items():
    print("%-8s gpu %.2f ms   host-enqueue %.2f ms" % (name, np.median([a for a, _ in v]), np.median([b for _, b in v])))

# sub-parts of the D step
from confignet_amd.losses import compute_discriminator_loss
real, fake = m.get_discriminator_batch(ds)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("get_discriminator_batch (encoder fwd + G fwd): %.2f ms" % timed(lambda: m.get_discriminator_batch(ds)))
def dloss():
    m.discriminator.zero_grad()
    l = compute_discriminator_loss(m.discriminator, real, fake)
    torch.autograd.backward(l["loss_sum"], inputs=m.discriminator.trainable_weights)
print("D loss fwd+bwd incl. R1: %.2f ms" % timed(dloss))
def dfwd():
    with torch.no_grad():
        m.discriminator(real); m.discriminator(fake)
print("2x D forward only: %.2f ms" % timed(dfwd))
def dnoR1():
    m.discriminator.zero_grad()
    o = m.discriminator(real); o2 = m.discriminator(fake)
    l = sum(v.sum() for v in o.values()) + sum(v.sum() for v in o2.values())
    torch.autograd.backward(l, inputs=m.discriminator.trainable_weights)
print("D fwd+bwd without R1 (fused path): %.2f ms" % timed(dnoR1))
