"""Where does an iteration's wall time go?  Host timestamps + device events around the phases of training_iteration."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ConfigNet, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs
np.random.seed(0)
ds = SyntheticFaceDataset(512, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 16, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0); m.use_graphs = True
m.setup_training(None, ds, 0, real_training_set=ds)
dopt, gopt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
for _ in range(5):
    m.training_iteration(ds, ds, dopt, gopt)
torch.cuda.synchronize()
# pipelined: K iterations, sync at the end only
K = 20
t0 = time.perf_counter()
host = []
for _ in range(K):
    h0 = time.perf_counter()
    m.training_iteration(ds, ds, dopt, gopt)
    host.append(time.perf_counter() - h0)
t_host_done = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("pipelined: %.2f ms per iteration wall; host returned after %.2f ms per iteration (median host time in training_iteration %.2f ms)"
      % (1e3 * t_all / K, 1e3 * t_host_done / K, 1e3 * np.median(host)))
# host-side split of one iteration
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m.training_iteration(ds, ds, dopt, gopt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative")
import io
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(22)
print(buf.getvalue()[:4500])
