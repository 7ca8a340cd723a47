"""N > 1 path on CPU: world_size-2 gloo processes exercise confignet_amd.parallel (the same code the
GPU ranks run over RCCL) and check the data-parallel identity the design relies on: averaging the
per-rank gradients of mean-reduced losses over equal shards == the gradient on the global batch
(including the per-sample R1 penalty), using the oracle's discriminator loss as the model."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import importlib
    parallel = importlib.import_module("confignet_amd.parallel")
    from oracle import ref_nets as R
    from oracle import ref_ops as O
    from oracle import ref_steps as S
    assert parallel.init_from_env() == world and parallel.rank() == rank and parallel.world_size() == world
    torch.set_num_threads(2)
    rng = np.random.default_rng(0)                      # identical weights on every rank
    shapes = R.discriminator_weight_shapes(64)
    w = [torch.tensor(O.glorot_uniform(rng, s) if len(s) > 1 else rng.normal(size=s) * 0.1, dtype=torch.float64,
                      requires_grad=True) for s in shapes]
    data = np.random.default_rng(1)
    real = torch.tensor(data.uniform(-1, 1, size=(4, 64, 64, 3)))
    fake = torch.tensor(data.uniform(-1, 1, size=(4, 64, 64, 3)))

    def loss_fn(r, f):
        real_in = r.detach().requires_grad_(True)
        o_r = R.discriminator_forward(w, real_in)
        o_f = R.discriminator_forward(w, f)
        total = 0
        for o in o_r.values():
            total = total + O.gan_d_loss(torch.ones_like(o), o) + O.r1_penalty(o, real_in)
        for o in o_f.values():
            total = total + O.gan_d_loss(torch.zeros_like(o), o)
        return total

    shard = slice(rank * 2, rank * 2 + 2)
    g_local = S.grads_of(loss_fn(real[shard], fake[shard]), w)
    flat = torch.cat([g.reshape(-1) for g in g_local]).contiguous()
    parallel.allreduce_flat_([flat])                    # mean over ranks, in place
    g_full = torch.cat([g.reshape(-1) for g in S.grads_of(loss_fn(real, fake), w)])
    err = float((flat - g_full).abs().max() / g_full.abs().max())
    # broadcast_weights makes rank 1 adopt rank 0's values
    class _N:
        def mark_updated(self):
            self.updated = True

        def non_trainable_changed(self):
            self.stats_dropped = True
    net = _N()
    net.arena = torch.full((8,), float(rank))
    net.weights = []
    parallel.broadcast_weights([net])
    ok_bcast = bool((net.arena == 0).all()) and net.updated and net.stats_dropped   # derived caches are told
    # optim.Adam.apply_gradients over two "networks" with one arena exchanged early (parallel.begin_allreduce, the overlapped
    # generator / regressor buckets of the data-parallel generator step) and one at apply time: every rank must end with the
    # SAME weights = one Keras-Adam step on the rank-mean gradient.  (The Adam launch itself is a HIP kernel: replaced by its
    # formula here, the ordering / exchange logic is what runs.)
    from confignet_amd import ops, optim

    def adam_step_cpu(theta, grad, m, v, ema, lr_t, b1, b2, eps, ema_alpha=0.999):
        m.mul_(b1).add_(grad, alpha=1 - b1)
        v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
        theta.sub_(lr_t * m / (v.sqrt() + eps))
    ops.adam_step = adam_step_cpu

    class _Net(_N):
        def __init__(self, n, seed):
            g = torch.Generator().manual_seed(seed)
            self.arena = torch.randn(n, generator=g)                                   # identical on every rank
            self.grad_arena = torch.randn(n, generator=torch.Generator().manual_seed(seed + 100 * (rank + 1)))   # per rank
    early, late = _Net(64, 1), _Net(32, 2)
    want = []
    for net in (early, late):
        gs = [torch.randn(net.arena.numel(), generator=torch.Generator().manual_seed((1 if net is early else 2) + 100 * (r + 1)))
              for r in range(world)]
        gm = sum(gs) / world
        want.append(net.arena - 4e-4 * (0.1 ** 0.5) * gm / ((0.1 * gm * gm).sqrt() + 1e-7))    # t = 1, beta_1 = 0, beta_2 = 0.9
    opt = optim.Adam(lr=4e-4, beta_1=0.0, beta_2=0.9)
    with optim.deferred_updates() as items:                       # what a captured step records ...
        opt.apply_gradients([early, late], advance=False, slot="g")
    assert len(items) == 1 and torch.equal(early.arena, _Net(64, 1).arena)      # nothing applied yet
    opt.advance("g")
    parallel.begin_allreduce([early])                             # ... the cut inside the step starts the early bucket ...
    optim.run_deferred(items)                                     # ... and StepGraph.finish() waits for it, reduces the rest, applies
    ok_adam = all(float((n.arena - w).abs().max()) < 1e-6 for n, w in zip((early, late), want)) and not parallel._pending
    # Global batch statistics of the normalised latent regression (confignet_second_stage.py:93-107, option (i) of SURVEY.md 8e):
    # with GlobalBatchMoments the rank-mean of the per-shard gradients equals the gradient of the single-process loss on the
    # whole batch, although every rank's loss depends on every rank's rows through the batch mean / variance.
    from confignet_amd.losses import normalized_latent_regression
    gen = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 6, generator=gen, dtype=torch.float64), torch.randn(8, 6, generator=gen, dtype=torch.float64)
    t1 = torch.randn(6, 7, generator=gen, dtype=torch.float64).requires_grad_(True)
    t2 = torch.randn(6, 7, generator=gen, dtype=torch.float64).requires_grad_(True)
    rows = slice(rank * 4, rank * 4 + 4)
    l_loc = normalized_latent_regression(X[rows] @ t1, Y[rows] @ t2, 10.0, global_statistics=True)
    g_loc = torch.cat([g.reshape(-1) for g in torch.autograd.grad(l_loc, [t1, t2])]).contiguous()
    parallel.allreduce_flat_([g_loc])
    l_all = normalized_latent_regression(X @ t1, Y @ t2, 10.0, global_statistics=False)
    g_all = torch.cat([g.reshape(-1) for g in torch.autograd.grad(l_all, [t1, t2])])
    l_mean = l_loc.detach().clone().reshape(1)
    parallel.allreduce_flat_([l_mean])
    err_stats = max(float((g_loc - g_all).abs().max() / g_all.abs().max()), abs(float(l_mean) - float(l_all)) / abs(float(l_all)))
    # the form the captured steps use (collectives on the calling thread: DeferredGlobalStatsRegression) gives the same gradient
    from confignet_amd.losses import DeferredGlobalStatsRegression
    o_loc, y_loc = X[rows] @ t1, Y[rows] @ t2
    term = DeferredGlobalStatsRegression(o_loc, y_loc, 10.0)
    pairs = term.cotangents()
    g_def = torch.cat([g.reshape(-1) for g in torch.autograd.grad([t for t, _ in pairs], [t1, t2], grad_outputs=[c for _, c in pairs])]).contiguous()
    parallel.allreduce_flat_([g_def])
    err_stats = max(err_stats, float((g_def - g_all).abs().max() / g_all.abs().max()), abs(float(term.value) - float(l_loc)))
    # ... and per-rank statistics (option (ii), the default) are NOT that objective
    l_ii = normalized_latent_regression(X[rows] @ t1, Y[rows] @ t2, 10.0, global_statistics=False)
    differs = abs(float(l_ii) - float(l_loc)) > 1e-6
    torch.save({"err": err, "bcast": ok_bcast, "adam": ok_adam, "err_stats": err_stats, "differs": differs, "theta": torch.cat([early.arena, late.arena])},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradient_identity_and_collectives_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert res["err"] < 1e-10, "DP-averaged gradient != global-batch gradient (%.3e)" % res["err"]
        assert res["bcast"] and res["adam"]
        assert res["err_stats"] < 1e-10 and res["differs"], "global batch statistics: DP objective != global-batch objective (%.3e)" % res["err_stats"]
    a, b = (torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))["theta"] for r in range(2))
    assert torch.equal(a, b), "replicated Adam must give bit-identical weights on every rank"
