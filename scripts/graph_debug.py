import sys, numpy as np, torch
sys.path.insert(0, ".")
from confignet_amd import ConfigNet, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs
ds = SyntheticFaceDataset(16, 128, seed=3)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "output_shape": (128, 128, 3)})
ds.process_metadata(cfg, True)
np.random.seed(5)
m = ConfigNet(cfg, seed=0)
m.use_graphs = True
m.setup_training(None, ds, 0, real_training_set=ds)
dopt, gopt = optim.Adam(**m.config["optimizer"]), optim.Adam(**m.config["optimizer"])
for it in range(3):
    out = m.training_iteration(ds, ds, dopt, gopt)
    print(it, [round(float(d["loss_sum"]), 3) for d in out], float(out[0]["GAN_loss_real_0"]))
nets = m.all_networks()
torch.cuda.synchronize()
snap = [n.arena.clone() for n in nets]
st = [{k: (a.clone(), b.clone()) for k, (a, b) in o._state.items()} for o in (dopt, gopt)]
out = m.training_iteration(ds, ds, dopt, gopt)
print("after snapshot", [round(float(d["loss_sum"]), 3) for d in out], float(out[0]["GAN_loss_real_0"]))
out = m.training_iteration(ds, ds, dopt, gopt)
print("next", [round(float(d["loss_sum"]), 3) for d in out], float(out[0]["GAN_loss_real_0"]))
# restore and run eagerly from the same state as "after snapshot"
def restore():
    for n, a in zip(nets, snap):
        n.arena.copy_(a); n.mark_updated()
    for o, s_ in zip((dopt, gopt), st):
        for k, (a, b) in s_.items():
            o._state[k][0].copy_(a); o._state[k][1].copy_(b)
rng = np.random.get_state()
it_d, it_g = dopt.iterations, gopt.iterations
snap = [n.arena.clone() for n in nets]
st = [{k: (a.clone(), b.clone()) for k, (a, b) in o._state.items()} for o in (dopt, gopt)]
outg = [{k: float(v) for k, v in d.items()} for d in m.training_iteration(ds, ds, dopt, gopt)]
restore(); np.random.set_state(rng); dopt.iterations, gopt.iterations = it_d, it_g
m.use_graphs = False
oute = [{k: float(v) for k, v in d.items()} for d in m.training_iteration(ds, ds, dopt, gopt)]
for name, a, b in zip("d sd ld g".split(), outg, oute):
    for k in a:
        flag = "" if abs(a[k] - b[k]) <= 1e-4 * max(1, abs(b[k])) else "   <<<<<"
        print(name, k, a[k], b[k], flag)
