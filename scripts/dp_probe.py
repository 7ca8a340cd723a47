"""Probe of the data-parallel dispatch on one GPU (CN_FORCE_DP=1): host enqueue time vs device time per
iteration, for the variants of parallel.allreduce_flat_ selected by CN_AR_MODE."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CN_FORCE_DP", "1")
import numpy as np
import torch

from confignet_amd import ConfigNet, SyntheticFaceDataset, optim, parallel
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs

parallel.init_from_env()
torch.cuda.set_device(0)
res, batch = 256, 16
real_set, synth_set = SyntheticFaceDataset(64, res, seed=1), SyntheticFaceDataset(64, res, seed=2)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": batch, "output_shape": (res, res, 3)})
synth_set.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
m.setup_training(None, synth_set, 0, real_training_set=real_set)
d_opt, g_opt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
m.use_graphs = True
for _ in range(3):
    m.training_iteration(real_set, synth_set, d_opt, g_opt)
torch.cuda.synchronize()
acc = {"ar": 0.0, "adam": 0.0, "replay": 0.0, "n_ar": 0}
from confignet_amd import ops
import torch.distributed as dist
_variant = os.environ.get("CN_AR_VARIANT", "")
_cs = torch.cuda.Stream()
def _variant_ar(bufs):
    if _variant == "mul":
        for b in bufs: b.mul_(1.0)
    elif _variant == "dist":
        for b in bufs: dist.all_reduce(b)
    elif _variant == "dance":
        cur = torch.cuda.current_stream(); _cs.wait_stream(cur)
        with torch.cuda.stream(_cs):
            for b in bufs: b.mul_(1.0)
        cur.wait_stream(_cs)
    elif _variant == "dist_small":
        for b in bufs: dist.all_reduce(b[:1024])
if _variant:
    parallel.allreduce_flat_ = _variant_ar
_ar, _adam = parallel.allreduce_flat_, ops.adam_step
def ar(bufs):
    t = time.perf_counter(); _ar(bufs); acc["ar"] += time.perf_counter() - t; acc["n_ar"] += 1
def adam(*a, **k):
    t = time.perf_counter(); _adam(*a, **k); acc["adam"] += time.perf_counter() - t
parallel.allreduce_flat_, ops.adam_step = ar, adam
_rep = torch.cuda.CUDAGraph.replay
def rep(self):
    t = time.perf_counter(); _rep(self); acc["replay"] += time.perf_counter() - t
torch.cuda.CUDAGraph.replay = rep
host, total = [], []
for _ in range(8):
    t0 = time.perf_counter()
    m.training_iteration(real_set, synth_set, d_opt, g_opt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print("variant=%s mode=%s host enqueue %.2f ms, iteration %.2f ms" % (_variant, os.environ.get("CN_AR_MODE", "comm"), np.median(host), np.median(total)))
print({k: round(v * 1e3 / 8, 3) for k, v in acc.items()})
import torch.distributed as dist
dist.destroy_process_group()
