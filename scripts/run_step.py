"""Run one step function repeatedly (for rocprofv3 kernel traces of a single step)."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from confignet_amd import ConfigNet, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs
which, reps = sys.argv[1], int(sys.argv[2])
np.random.seed(0)
ds = SyntheticFaceDataset(64, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 16, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
m.setup_training(None, ds, 0, real_training_set=ds)
m.use_graphs = os.environ.get("CN_USE_GRAPHS") == "1"
dopt, gopt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
fn = {"d": lambda: m.discriminator_training_step(ds, dopt), "sd": lambda: m.synth_discriminator_training_step(ds, dopt),
      "ld": lambda: m.latent_discriminator_training_step(ds, ds, dopt), "g": lambda: m.generator_training_step(ds, ds, gopt)}[which]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
