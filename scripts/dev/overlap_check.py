"""Cross-iteration overlap of the discriminator steps: same seeds, flag off vs on -- losses of every iteration and the final
weights must agree (the np.random draws happen in the same order; gradients of the two halves are added in the arena)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ConfigNet, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs
res = {}
MODES = [(0, False), (1, True)] if len(sys.argv) < 2 else [(0, False), (1, False)]
for slot, flag in MODES:
    np.random.seed(0)
    ds = SyntheticFaceDataset(64, 128, seed=1)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "output_shape": (128, 128, 3)})
    ds.process_metadata(cfg, True)
    m = ConfigNet(cfg, seed=0)
    m.setup_training(None, ds, 0, real_training_set=ds)
    m.use_graphs = True
    m.overlap_discriminators = flag
    dopt, gopt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
    hist = []
    log = []
    orig = m._stage_real
    def logged(key, dataset, n, orig=orig, log=log):
        st = np.random.get_state()
        idx = np.random.randint(0, dataset.imgs.shape[0], n)
        np.random.set_state(st)
        log.append((key, tuple(idx.tolist())))
        return orig(key, dataset, n)
    m._stage_real = logged
    for it in range(8):
        out = m.training_iteration(ds, ds, dopt, gopt)
        hist.append([{k: float(v) for k, v in d.items()} for d in out])
    torch.cuda.synchronize()
    res[slot + 2] = log
    res[slot] = (hist, [w.copy() for w in m.discriminator.get_weights()], [w.copy() for w in m.generator.get_weights()])
    print("flag", flag, "graphs:", {k[0]: (len(v.segments), v.early_cut) for k, v in m._graphs.items()})
print("same sequence of real-image draws (common prefix):", res[2] == res[3][:len(res[2])], len(res[2]), len(res[3]))
a1, b1 = res[0][0][1][0], res[1][0][1][0]
print("iteration 1, D dict:", {k: round(abs(a1[k] - b1[k]) / max(1.0, abs(a1[k])), 5) for k in a1})
worst = 0.0
for it, (a, b) in enumerate(zip(res[0][0], res[1][0])):
    for da, db in zip(a, b):
        for k in da:
            worst = max(worst, abs(da[k] - db[k]) / max(1.0, abs(da[k])))
    print("iter", it, "max rel loss deviation so far %.3e" % worst, " D loss_sum %.5f / %.5f  G loss_sum %.5f / %.5f" % (a[0]["loss_sum"], b[0]["loss_sum"], a[3]["loss_sum"], b[3]["loss_sum"]))
for name, i in (("D", 1), ("G", 2)):
    dev = max(float(np.abs(x - y).max()) for x, y in zip(res[0][i], res[1][i]))
    print(name, "weights max abs deviation %.3e" % dev)
