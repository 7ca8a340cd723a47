"""Helper of test_steps_gpu.py::test_two_rank_data_parallel_run_of_the_benchmarked_dispatch: the benchmark's own model
(bench.setup) at 256 x 256 in the benchmark's dispatch (step graphs, concurrent discriminator phase, cross-iteration overlap,
two-part generator backward with the early gradient buckets).

    rank mode:   one of the 2 ranks of a gloo group that SHARE the one device (CN_DP_BACKEND=gloo, CN_DP_SHARE_DEVICE=1):
                 batch 8 per rank, batches drawn from np.random.seed(seed + rank), every staged batch logged
    single mode: one process, batch 16, re-running the CONCATENATED batches of the two ranks (StaticBuffers.replay)

--dtype=bf16 anywhere on the command line: the same run with bf16 activations (ops.set_activation_dtype).

Both write: the loss scalars of three iterations, every (optimizer, network)'s Adam first moment after iteration 1
(= (1 - beta_1) x the gradient that Adam saw: the all-reduced rank mean resp. the global-batch gradient), final weights and
moments, and (rank mode) the staged batches."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, mode, batch, global_stats, deterministic, replay_paths, dtype="f32"):
    import bench
    from confignet_amd import ops, parallel
    ops.set_activation_dtype(dtype)                              # "bf16": BASELINE.json configs[2]'s compute type
    ops.set_deterministic(bool(deterministic))
    if mode == "rank":
        parallel.init_from_env()
        assert parallel.world_size() == 2 and parallel.active()
    rank = parallel.rank()
    torch.cuda.set_device(0)
    model, real_set, synth_set, d_opt, g_opt, cfg = bench.setup(batch, 256, 64, rank)
    model.config["dp_global_batch_statistics"] = bool(global_stats)
    model.use_graphs = True
    model.overlap_discriminators = True
    nets = model.all_networks()
    start = [n.get_weights() for n in nets]
    for _ in range(4):                                           # eager call, capture, first replays
        model.training_iteration(real_set, synth_set, d_opt, g_opt)
    torch.cuda.synchronize()
    assert len(model._graphs) == 4 and all(g.graph is not None for g in model._graphs.values())
    split = [bool(g.split) for g in model._graphs.values()]
    n_segments = {g.name: len(g.segments) for g in model._graphs.values()}
    for n, w0 in zip(nets, start):                               # back to the initial state; the captured graphs stay
        n.set_weights(w0)
    for o in (d_opt, g_opt):
        o.iterations = 0
        for mom, var in o._state.values():
            mom.zero_()
            var.zero_()
    np.random.seed(77 + rank)
    if mode == "rank":
        model._bufs.log = {}
    else:
        logs = [np.load(p) for p in replay_paths]
        keys = [k[4:] for k in logs[0].files if k.startswith("log/")]
        rep = {}
        for k in keys:
            per_rank = [l["log/" + k] for l in logs]             # (calls, per-rank batch, ...)
            assert all(a.shape == per_rank[0].shape for a in per_rank)
            rep[k] = [np.concatenate([a[i] for a in per_rank], axis=0) for i in range(per_rank[0].shape[0])]
        model._bufs.replay = rep

    def moments(tag):
        out, k = {}, 0
        for o in (d_opt, g_opt):
            for net in nets:
                st = o._state.get(id(net))
                if st is not None:
                    out["%sm%d" % (tag, k)], out["%sv%d" % (tag, k)] = st[0].cpu().numpy(), st[1].cpu().numpy()
                    k += 1
        return out

    losses, state = [], {}
    for it in range(3):
        out = model.training_iteration(real_set, synth_set, d_opt, g_opt)
        losses.append([float(v) for d in out for v in d.values()])
        if it == 0:
            torch.cuda.synchronize()
            state.update(moments("first_"))
    torch.cuda.synchronize()
    if mode == "single":
        left = {k: len(v) for k, v in model._bufs.replay.items() if v}
        # the pipelined loop stages the NEXT iteration's batches at the end of an iteration: every logged draw must have been consumed
        assert not left, left
    state.update({"w%d" % i: n.arena.detach().cpu().numpy() for i, n in enumerate(nets)})
    state.update(moments("last_"))
    if mode == "rank":
        for k, v in model._bufs.log.items():
            state["log/" + k] = np.stack(v)
    names = [k for d in out for k in d.keys()]
    np.savez(out_path, losses=np.array(losses), loss_names=np.array(names), dp=np.array([parallel.active()]), split=np.array(split),
             g_segments=np.array([n_segments.get("g", 0)]), beta_1=np.array([cfg["optimizer"].get("beta_1", 0.9)]), **state)
    if mode == "rank":
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if not a.startswith("--dtype=")]
    dt = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--dtype=")), "f32")
    main(argv[0], argv[1], int(argv[2]), int(argv[3]), int(argv[4]), argv[5:], dt)
