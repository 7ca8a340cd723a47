"""Data parallelism: one process per GPU, gradients summed with RCCL over xGMI.

Every network keeps its gradients in one contiguous arena, so the exchange is ONE all-reduce per
network per step (D: 10.7 MB, G step: generator 32 MB + latent regressor 30 MB + encoder 94 MB),
issued on a side stream as soon as the backward pass has been enqueued.  Replicated Adam state
gives identical updates on every rank, so no weight broadcast is needed after step 0."""
import os

import torch
import torch.distributed as dist

_comm_stream = None


def init_from_env():
    """Initialise torch.distributed from torchrun's environment (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, init_method="env://")
    return world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_flat_(buffers):
    """In-place mean over ranks of each flat buffer (sum all-reduce, then 1/world)."""
    ws = world_size()
    if ws == 1:
        return
    global _comm_stream
    if buffers[0].is_cuda:
        if _comm_stream is None:
            _comm_stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        _comm_stream.wait_stream(cur)
        with torch.cuda.stream(_comm_stream):
            works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True) for b in buffers]
            for w in works:
                w.wait()
            for b in buffers:
                b.mul_(1.0 / ws)
        cur.wait_stream(_comm_stream)
    else:
        for b in buffers:
            dist.all_reduce(b, op=dist.ReduceOp.SUM)
            b.mul_(1.0 / ws)


def allreduce_gradients(nets):
    if world_size() > 1:
        allreduce_flat_([n.grad_arena for n in nets])


def broadcast_weights(nets, src=0):
    """Make every rank start from rank `src`'s weights (used once after construction)."""
    if world_size() == 1:
        return
    for n in nets:
        dist.broadcast(n.arena, src=src)
        for w in n.weights:
            if not w.requires_grad:
                dist.broadcast(w, src=src)
