import sys, numpy as np, torch
sys.path.insert(0, ".")
from confignet_amd import ConfigNet, ConfigNetFirstStage, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs
res, b, stage = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
which = sys.argv[4] if len(sys.argv) > 4 else "all"
ds = SyntheticFaceDataset(16, res, seed=3)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": b, "output_shape": (res, res, 3)})
ds.process_metadata(cfg, True)
np.random.seed(5)
m = (ConfigNet if stage == "2" else ConfigNetFirstStage)(cfg, seed=0)
m.use_graphs = True
m.setup_training(None, ds, 0, real_training_set=ds)
dopt, gopt = optim.Adam(**m.config["optimizer"]), optim.Adam(**m.config["optimizer"])
for it in range(3):
    if which in ("all", "d"): m.discriminator_training_step(ds, dopt)
    if which in ("all", "sd"): m.synth_discriminator_training_step(ds, dopt)
    if which in ("all", "ld"): (m.latent_discriminator_training_step(ds, ds, dopt) if stage == "2" else m.latent_discriminator_training_step(ds, dopt))
    if which in ("all", "g"): g = m.generator_training_step(ds, ds, gopt)
    torch.cuda.synchronize()
print("OK", res, b, stage, which)
