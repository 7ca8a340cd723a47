"""GPU time of the concurrent discriminator phase vs the generator step in HIP-graph mode."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from confignet_amd import ConfigNet, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs
np.random.seed(0)
ds = SyntheticFaceDataset(64, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 16, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0); m.use_graphs = True
m.setup_training(None, ds, 0, real_training_set=ds)
dopt, gopt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
for _ in range(3):
    m.training_iteration(ds, ds, dopt, gopt)
torch.cuda.synchronize()
def timed(fn, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return np.median(ts)
steps = {"d": lambda: m.discriminator_training_step(ds, dopt), "sd": lambda: m.synth_discriminator_training_step(ds, dopt),
         "ld": lambda: m.latent_discriminator_training_step(ds, ds, dopt), "g": lambda: m.generator_training_step(ds, ds, gopt)}
for k, f in steps.items():
    print("%-3s alone (graph replay): %.2f ms" % (k, timed(f)))
print("d+sd+ld concurrent: %.2f ms" % timed(lambda: m.run_concurrently([steps["d"], steps["sd"], steps["ld"]])))
print("whole iteration: %.2f ms" % timed(lambda: m.training_iteration(ds, ds, dopt, gopt)))
