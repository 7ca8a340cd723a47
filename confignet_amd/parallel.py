"""Data parallelism: one process per GPU, gradients summed with RCCL over xGMI.

Every network keeps its gradients in one contiguous arena, so the exchange is ONE all-reduce per
network per step (D: 10.7 MB, G step: generator 32 MB + latent regressor 30 MB + encoder 94 MB),
issued right after the backward pass (after the replay of the step's captured graph).  Replicated Adam state
gives identical updates on every rank, so no weight broadcast is needed after step 0."""
import os

import torch
import torch.distributed as dist


def _forced():
    """CN_FORCE_DP=1 runs the multi-rank code path (process group, all-reduces, split step graphs) even with a
    single rank: the way the RCCL path is exercised on a one-GPU box."""
    return os.environ.get("CN_FORCE_DP", "0") == "1"


def init_from_env():
    """Initialise torch.distributed from torchrun's environment (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if dist.is_initialized() or (world <= 1 and not _forced()):
        return world
    if world <= 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, init_method="env://")
    return world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active():
    """True when gradients have to be exchanged (more than one rank, or CN_FORCE_DP=1 with a process group)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_flat_(buffers):
    """In-place mean over ranks of each flat buffer.  On the GPU: one RCCL all-reduce (ncclAvg) per arena, issued
    from the calling stream -- torch orders its RCCL stream after that stream and the caller's later work after the
    collective, so there is no extra scaling pass and no stream of our own (measured on one GPU with CN_FORCE_DP=1:
    a hand-rolled side-stream + 1/world pass cost 12 ms per iteration in cross-stream hops, this form costs none)."""
    if not active():
        return
    if buffers[0].is_cuda:
        for b in buffers:
            dist.all_reduce(b, op=dist.ReduceOp.AVG)
    else:                                          # gloo (CPU tests): no AVG
        ws = world_size()
        for b in buffers:
            dist.all_reduce(b, op=dist.ReduceOp.SUM)
            b.mul_(1.0 / ws)


def allreduce_sum_inline(t):
    """SUM over ranks of a small tensor IN THE MIDDLE of a step (global batch statistics of a loss): eager dispatch reduces on
    the spot; under HIP-graph capture the step's graph is cut here and the collective runs between the two segments at every
    replay (graphs.segment_break), on the tensor's fixed address in the graph's pool."""
    if not active():
        return t
    from . import graphs
    graphs.segment_break(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM))
    return t


_pending = {}        # id(net) -> async work handle of an all-reduce of its gradient arena that is in flight


def begin_allreduce(nets):
    """Start the mean all-reduce of these networks' gradient arenas WITHOUT waiting for it: RCCL runs it on torch's
    communication stream, ordered after everything already queued on the calling stream, while the calling stream goes on
    with compute (the rest of the backward pass).  `allreduce_gradients` later waits for it instead of reducing again."""
    if not active():
        return
    ws = world_size()
    for n in nets:
        assert id(n) not in _pending, "gradient all-reduce already in flight"
        if n.grad_arena.is_cuda:
            _pending[id(n)] = (dist.all_reduce(n.grad_arena, op=dist.ReduceOp.AVG, async_op=True), None)
        else:                                      # gloo (CPU tests): no AVG
            _pending[id(n)] = (dist.all_reduce(n.grad_arena, op=dist.ReduceOp.SUM, async_op=True), 1.0 / ws)


def allreduce_gradients(nets):
    """Mean over ranks of every network's gradient arena: waits for the ones `begin_allreduce` already started (the calling
    stream is ordered after the collective), reduces the others now."""
    if not active():
        return
    rest = []
    for n in nets:
        pend = _pending.pop(id(n), None)
        if pend is None:
            rest.append(n.grad_arena)
        else:
            work, scale = pend
            work.wait()
            if scale is not None:
                n.grad_arena.mul_(scale)
    if rest:
        allreduce_flat_(rest)


def broadcast_weights(nets, src=0):
    """Make every rank start from rank `src`'s weights (used once after construction)."""
    if world_size() == 1:
        return
    for n in nets:
        dist.broadcast(n.arena, src=src)
        for w in n.weights:
            if not w.requires_grad:
                dist.broadcast(w, src=src)
        n.mark_updated()
        n.non_trainable_changed()
