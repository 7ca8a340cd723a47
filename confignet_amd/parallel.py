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
    backend = os.environ.get("CN_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    assert backend in ("nccl", "gloo"), "CN_DP_BACKEND must be nccl or gloo"
    if torch.cuda.is_available():
        # CN_DP_SHARE_DEVICE=1 (with CN_DP_BACKEND=gloo): the ranks share the visible devices round-robin -- the way N > 1
        # ranks of the product path run on a one-GPU box (RCCL refuses two ranks on one device)
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("CN_DP_SHARE_DEVICE", "0") == "1":
            assert backend == "gloo", "CN_DP_SHARE_DEVICE=1 needs CN_DP_BACKEND=gloo (RCCL wants one device per rank)"
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, init_method="env://")
    return world


def _host_staged(t):
    """True when the collective on `t` goes through a host copy: a device tensor on a gloo group (CN_DP_BACKEND=gloo, the
    N-ranks-on-one-GPU test configuration).  The copy down is issued on the calling stream and waited for, the copy back is
    issued on the calling stream again: the same ordering against the caller's launches as RCCL's stream-ordered form."""
    return t.is_cuda and dist.get_backend() == "gloo"


class _HostWork:
    """An in-flight SUM all-reduce of a device tensor through its host copy (gloo); wait() writes the result back (scaled)."""

    def __init__(self, t, scale, async_op):
        self.t, self.scale = t, scale
        self.host = t.detach().to("cpu")                      # waits for the calling stream up to here
        self.work = dist.all_reduce(self.host, op=dist.ReduceOp.SUM, async_op=True)
        if not async_op:
            self.wait()

    def wait(self):
        self.work.wait()
        if self.scale != 1.0:
            self.host.mul_(self.scale)
        self.t.copy_(self.host)                               # on the calling stream


def _all_reduce_mean(t, async_op=False):
    """Mean over ranks of `t`, in place.  Returns None, or with async_op an object with wait() that the caller MUST call
    before it reads `t` (for RCCL the wait orders the calling stream after the collective; no host blocking)."""
    ws = world_size()
    if _host_staged(t):
        w = _HostWork(t, 1.0 / ws, async_op)
        return w if async_op else None
    if t.is_cuda:
        # ncclAvg: no scaling pass.  ALWAYS issued as an asynchronous work (and waited for at once when the caller wants it in
        # order): torch runs a synchronous collective on the CALLING stream, and the end event it records there is polled by the
        # process group's watchdog thread -- if that stream has entered a HIP-graph capture by the time of the poll, hipEventQuery
        # fails with hipErrorCapturedEvent and the watchdog takes the process down (seen twice in the forced-DP test, round 6).
        # An asynchronous work runs on torch's own RCCL stream, which never captures; wait() orders the calling stream after it.
        w = dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)
        if async_op:
            return w
        w.wait()
        return None

    class _Scaled:                                  # gloo on host tensors (CPU tests): no AVG
        def __init__(self, work):
            self.work = work

        def wait(self):
            self.work.wait()
            t.mul_(1.0 / ws)
    w = _Scaled(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))
    if async_op:
        return w
    w.wait()
    return None


def all_reduce_sum(t):
    """SUM over ranks of `t`, in place, ordered against the calling stream."""
    if _host_staged(t):
        _HostWork(t, 1.0, False)
    elif t.is_cuda:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True).wait()        # (on torch's RCCL stream: see _all_reduce_mean)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_max(t):
    if _host_staged(t):
        h = t.detach().to("cpu")
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
    elif t.is_cuda:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True).wait()
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def broadcast_(t, src=0):
    if _host_staged(t):
        h = t.detach().to("cpu")
        dist.broadcast(h, src=src)
        t.copy_(h)
    elif t.is_cuda:
        dist.broadcast(t, src=src, async_op=True).wait()
    else:
        dist.broadcast(t, src=src)
    return t


def barrier():
    """All ranks reach this point (host-blocking).  On RCCL as an asynchronous one-element all-reduce on torch's own RCCL stream
    (see _all_reduce_mean: a synchronous collective would leave an end event on the calling stream for the watchdog to poll after
    that stream may have started a HIP-graph capture)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_backend() == "nccl" and torch.cuda.is_available():
        t = torch.zeros(1, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True).wait()
        torch.cuda.current_stream().synchronize()
    else:
        dist.barrier()


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
    for b in buffers:
        _all_reduce_mean(b)


def allreduce_sum_inline(t):
    """SUM over ranks of a small tensor IN THE MIDDLE of a step (global batch statistics of a loss): eager dispatch reduces on
    the spot; under HIP-graph capture the step's graph is cut here and the collective runs between the two segments at every
    replay (graphs.segment_break), on the tensor's fixed address in the graph's pool."""
    if not active():
        return t
    from . import graphs
    graphs.segment_break(lambda: all_reduce_sum(t))
    return t


_pending = {}        # id(net) -> async work handle of an all-reduce of its gradient arena that is in flight


def begin_allreduce(nets):
    """Start the mean all-reduce of these networks' gradient arenas WITHOUT waiting for it: RCCL runs it on torch's
    communication stream, ordered after everything already queued on the calling stream, while the calling stream goes on
    with compute (the rest of the backward pass).  `allreduce_gradients` later waits for it instead of reducing again."""
    if not active():
        return
    for n in nets:
        assert id(n) not in _pending, "gradient all-reduce already in flight"
        _pending[id(n)] = _all_reduce_mean(n.grad_arena, async_op=True)


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
            pend.wait()
    if rest:
        allreduce_flat_(rest)


def broadcast_weights(nets, src=0):
    """Make every rank start from rank `src`'s weights (used once after construction)."""
    if world_size() == 1:
        return
    for n in nets:
        broadcast_(n.arena, src=src)
        for w in n.weights:
            if not w.requires_grad:
                broadcast_(w, src=src)
        n.mark_updated()
        n.non_trainable_changed()
