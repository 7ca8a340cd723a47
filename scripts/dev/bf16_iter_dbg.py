import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from confignet_amd import ConfigNet, SyntheticFaceDataset, ops, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs
run_metrics = sys.argv[1] == "1"
ds = SyntheticFaceDataset(16, 128, seed=3)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "output_shape": (128, 128, 3), "run_metrics": run_metrics})
ds.process_metadata(cfg, True)
out = {}
for mode in ("f32", "bf16", "f32"):
    ops.set_activation_dtype(mode)
    np.random.seed(5)
    m = ConfigNet(cfg, seed=0)
    m.setup_training(None, ds, 0, real_training_set=ds)
    dopt, gopt = optim.Adam(**m.config["optimizer"]), optim.Adam(**m.config["optimizer"])
    r = [{k: float(v) for k, v in d.items()} for d in m.training_iteration(ds, ds, dopt, gopt)]
    print(mode, {k: round(v, 4) for k, v in r[0].items()})
