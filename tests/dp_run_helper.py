"""Helper of test_steps_gpu.py::test_data_parallel_dispatch_with_overlap_matches_single_process: runs a few second-stage
iterations in the benchmark's dispatch (graphs + cross-iteration overlap) in deterministic mode and writes the final weights,
Adam moments and loss scalars.  Run once plainly and once with CN_FORCE_DP=1 (1-rank RCCL group: split step graphs, eager
all-reduce + Adam after every replay, the two-part generator backward with its early all-reduce, global batch statistics)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, global_stats):
    from confignet_amd import ConfigNet, SyntheticFaceDataset, ops, optim, parallel
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.confignet_utils import merge_configs
    ops.set_deterministic(True)
    parallel.init_from_env()
    res, batch = 128, 4
    real_set, synth_set = SyntheticFaceDataset(8, res, seed=5), SyntheticFaceDataset(8, res, seed=6)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": batch, "output_shape": (res, res, 3), "dp_global_batch_statistics": bool(global_stats)})
    synth_set.process_metadata(cfg, True)
    np.random.seed(11)
    m = ConfigNet(cfg, seed=12)
    m.setup_training(None, synth_set, 0, real_training_set=real_set)
    d_opt, g_opt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
    m.use_graphs = True
    m.overlap_discriminators = True
    nets = m.all_networks()
    start = [n.get_weights() for n in nets]
    for _ in range(4):
        m.training_iteration(real_set, synth_set, d_opt, g_opt)
    torch.cuda.synchronize()
    assert all(g.graph is not None for g in m._graphs.values())
    split = [bool(g.split) for g in m._graphs.values()]
    # back to the initial state (the captured graphs stay): the three iterations below start from identical weights in every mode
    for n, w0 in zip(nets, start):
        n.set_weights(w0)
    for o in (d_opt, g_opt):
        o.iterations = 0
        for mom, var in o._state.values():
            mom.zero_()
            var.zero_()
    losses = []
    np.random.seed(77)
    first = None
    for it in range(3):
        out = m.training_iteration(real_set, synth_set, d_opt, g_opt)
        losses.append([float(v) for d in out for v in d.values()])
        if it == 0:
            first = {"f%d" % i: n.arena.detach().cpu().numpy() for i, n in enumerate(nets)}
    torch.cuda.synchronize()
    state = {"w%d" % i: n.arena.detach().cpu().numpy() for i, n in enumerate(nets)}
    state.update(first)
    k = 0
    for o in (d_opt, g_opt):
        for net in nets:
            st = o._state.get(id(net))
            if st is not None:
                state["m%d" % k], state["v%d" % k] = st[0].cpu().numpy(), st[1].cpu().numpy()
                k += 1
    np.savez(out_path, losses=np.array(losses), dp=np.array([parallel.active()]), split=np.array(split), **state)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
