"""HIP-graph capture of whole training steps.

A reference training step issues ~10^3 small ops (the six R1 input-gradient passes alone are hundreds of
launches); dispatched eagerly from Python the discriminator steps are host-bound (step_breakdown: GPU time ==
enqueue time).  Each step function is therefore split into a host half (numpy batch sampling, upload into
STATIC device buffers, optimizer.advance()) and a device half that only reads those buffers; the device half
(forward, R1 tape, backward, Adam) is captured once into a HIP graph and replayed."""
import torch

from . import ops
from . import optim
from . import parallel
from .nn import WEIGHTS_EPOCH


_recording = None        # the StepGraph whose fn is being captured right now


def segment_break(action=None, early=False):
    """Called by a step function between two parts of its device half; under capture the step's graph is cut here (replay =
    segment, action, segment, ...), eager dispatch just runs `action` on the spot.
    * `action()`: an eager launch that cannot live inside a HIP graph (an RCCL all-reduce) and that should start as soon as the
      first part has run while the second part goes on (data-parallel generator step: the generator / regressor gradient
      arenas are exchanged under the encoder's backward).
    * early=True marks that everything BEFORE this point reads nothing the sibling steps of the iteration write (the
      generator step's generator / encoder / VGG forward passes do not touch discriminator weights): the iteration replays
      those segments next to the discriminator-type graphs instead of after them (ConfigNetFirstStage._flush_deferred)."""
    g = _recording
    if g is None:
        if action is not None:
            action()
    else:
        g._cut(action)
        if early:
            g.early_cut = len(g.segments)


def _release_det_stream(stream):
    from ._lib import lib
    lib.cn_det_release_stream(stream.cuda_stream)


class StepGraph:
    """fn() -> dict of detached device scalars.  Call 1..warmup run eagerly ON THE CAPTURE STREAM (autograd's
    per-leaf AccumulateGrad nodes are bound to the stream they are created on, so the leaves must first be
    used under the stream the capture will use); the next call captures; later calls replay.

    Data-parallel runs (parallel.active()): the graph holds forward + backward only; the optimizer calls made by
    fn are recorded (optim.deferred_updates) and issued eagerly after every replay by `finish()` -- gradient
    all-reduce over RCCL, then Adam -- on whatever stream the replay was issued on.  fn may also cut itself into several
    graph segments with eager launches in between (segment_break)."""

    def __init__(self, fn, warmup=1, stream=None):
        self.fn, self.warmup = fn, warmup
        self.calls, self.graph, self.out = 0, None, None
        self.stream = stream if stream is not None else torch.cuda.Stream()
        if stream is None:
            # a stream of our own: in deterministic mode the library pins a per-stream workspace to it while a graph captured on it
            # may replay (cn_det_ws); hand the slot back when this StepGraph (and with it its graphs) is dropped
            import weakref
            weakref.finalize(self, _release_det_stream, self.stream)
        self.split = parallel.active()
        self.tail = []
        self.packed, self._result = None, None      # all loss scalars of the step in one static tensor / this call's copy
        self.segments = []           # [(CUDAGraph, action run after it | None)]; self.graph is the last segment
        self.early_cut = 0           # segments[:early_cut] may run concurrently with the iteration's sibling steps
        self.updated_nets = []       # networks whose Adam launches are nodes of this graph: every replay changes their weights

    def _run_fn(self):
        if not self.split:
            return self.fn()
        with optim.deferred_updates() as items:
            out = self.fn()
        self.tail = list(items)
        return out

    def finish(self):
        """Eager tail of one execution: the optimizer calls of a data-parallel step (no-op for single-rank graphs, whose Adam
        launches are graph nodes), then the copy of the step's loss scalars out of the graph's static buffers into the
        tensor handed to the caller for THIS execution (one small launch)."""
        if self.tail:
            optim.run_deferred(self.tail)
        for net in self.updated_nets:      # graph-node Adam launches do not run Python: eager forwards between replays must not
            net.mark_updated()             # find derived filter copies / inference graphs of the previous weights (nn.Net.epoch)
        if self._result is not None and self.packed is not None:
            self._result.copy_(self.packed)
            # allocated by result() on the CALLER's stream, written here on the stream the replay was issued on: the caching
            # allocator must not hand the block to the caller's stream again (once the caller drops the dict) before this copy ran
            self._result.record_stream(torch.cuda.current_stream())
            self._result = None

    def result(self):
        """The loss dict of the NEXT execution: scalars that are views of a fresh tensor, filled by `finish()` right after
        the replay -- so a caller may keep last step's dict while the next step runs (the graph's own outputs are static
        buffers that every replay overwrites)."""
        if self.packed is None:
            return self.out
        self._result = torch.empty_like(self.packed)
        return {k: self._result[i] for i, k in enumerate(self.out.keys())}

    # RCCL's watchdog thread polls events while a process group exists: keep its calls out of the capture
    def _mode(self):
        return "thread_local" if self.split else "global"

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        pool = self.segments[0][0].pool() if self.segments else None
        if pool is not None:
            g.capture_begin(pool=pool, capture_error_mode=self._mode())
        else:
            g.capture_begin(capture_error_mode=self._mode())
        self._cur = g

    def _cut(self, action):
        self._cur.capture_end()
        self.segments.append((self._cur, action))
        self._begin()

    def replay(self, start=0, stop=None):
        for g, action in self.segments[start:stop]:
            g.replay()
            if action is not None:
                action()

    def __call__(self):
        global _recording
        if self.graph is None:
            cur = torch.cuda.current_stream()
            if self.calls < self.warmup:
                self.calls += 1
                self.stream.wait_stream(cur)
                with torch.cuda.stream(self.stream):
                    out = self._run_fn()
                cur.wait_stream(self.stream)
                self.finish()
                return out
            torch.cuda.synchronize()
            # The cyclic garbage collector must not run inside a capture: if it frees an old graph / tensor of another pool
            # there, the runtime aborts the process (seen as "Fatal Python error: Aborted ... Garbage-collecting" in the
            # middle of a step's capture).  Collect now, keep it off until the capture has ended.
            import gc
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            ops.prof_enable(False)                 # no event records inside a capture
            WEIGHTS_EPOCH[0] += 1                  # derived caches must be rebuilt INSIDE this graph
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self._begin()
                _recording = self
                optim._touched = self.updated_nets = []
                try:
                    self.out = self._run_fn()
                    if isinstance(self.out, dict) and self.out and all(torch.is_tensor(v) and v.numel() == 1 for v in self.out.values()):
                        self.packed = torch.stack([v.reshape(()).float() for v in self.out.values()])
                finally:
                    _recording = None
                    optim._touched = None
                    self._cur.capture_end()
                    if gc_was_on:
                        gc.enable()
                self.segments.append((self._cur, None))
            cur.wait_stream(self.stream)
            self.graph = self._cur
            WEIGHTS_EPOCH[0] += 1
        res = self.result()
        self.replay()
        self.finish()
        return res


class InferenceGraph:
    """A forward pass without a tape, captured once per input signature and replayed: the demo / predict path issues ~60
    launches for ~1 ms of work at N = 1 (evaluation/confignet_demo.py:154-165 calls it every frame), so its latency is the
    dispatch.  `fn(**inputs)` reads only the static input tensors and network weights (at their fixed arena addresses: new
    weight VALUES are picked up by the replay; derived filter copies are made before the capture and the graph is dropped
    when the network's epoch moves).  Returns fn's output tensor (static: copy it out before the next replay)."""

    def __init__(self, fn, inputs, stream):
        self.inputs = {k: v.clone() for k, v in inputs.items()}
        self.stream = stream
        self._pinned = {}
        cur = torch.cuda.current_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream), torch.no_grad():
            fn(**self.inputs)                                   # warm-up: allocator, derived filter copies
            torch.cuda.current_stream().synchronize()
            self.graph = torch.cuda.CUDAGraph()
            mode = "thread_local" if parallel.active() else "global"
            ops._keepalive = self.derived = []                  # the graph reads these copies: they live as long as it does
            import gc
            gc.collect()                                        # (no garbage collection inside a capture: see StepGraph)
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                self.graph.capture_begin(capture_error_mode=mode)
                self.out = fn(**self.inputs)
                self.graph.capture_end()
            finally:
                ops._keepalive = None
                if gc_was_on:
                    gc.enable()
        cur.wait_stream(stream)

    def __call__(self, **inputs):
        """inputs: device tensors, or host arrays / tensors (staged through pinned mirrors, one upload per distinct object:
        the five AdaIN latents of the generator are normally the same array)."""
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            uploaded = {}
            for k, v in inputs.items():
                dst = self.inputs[k]
                if torch.is_tensor(v) and v.is_cuda:
                    dst.copy_(v, non_blocking=True)
                    continue
                src = uploaded.get(id(v))
                if src is None:
                    pin = self._pinned.get(k)
                    if pin is None:
                        pin = self._pinned[k] = torch.empty(dst.shape, dtype=dst.dtype, pin_memory=True)
                    pin.copy_(torch.as_tensor(v).reshape(dst.shape))
                    dst.copy_(pin, non_blocking=True)
                    uploaded[id(v)] = dst
                else:
                    dst.copy_(src, non_blocking=True)
            self.graph.replay()
        cur.wait_stream(self.stream)
        return self.out


def independent_streams(n, candidates=12, spin_us=300.0):
    """`n` streams that provably run side by side.  HIP streams share a handful of hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default) and two streams on one queue execute back to back; which streams collide depends
    on everything the process created before, so it is measured: a one-wave kernel that busy-waits `spin_us` is
    launched on a pair of candidate streams and the pair is kept apart if the two launches took about twice as long
    as one.  Greedy selection over `candidates` fresh streams; falls back to colliding streams when fewer than `n`
    independent ones exist."""
    import time
    from ._lib import lib
    ticks = int(spin_us * 100)                      # wall_clock64 runs at 100 MHz
    cands = [torch.cuda.Stream() for _ in range(candidates)]

    def pair_ms(a, b):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.check(lib.cn_spin(ticks, a.cuda_stream), "cn_spin")
        ops.check(lib.cn_spin(ticks, b.cuda_stream), "cn_spin")
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for st in cands[:2]:                            # warm the launch path
        pair_ms(st, st)
    serial = min(pair_ms(cands[0], cands[0]) for _ in range(3))          # two spins on ONE stream
    chosen = [cands[0]]
    for c in cands[1:]:
        if len(chosen) == n:
            break
        if all(min(pair_ms(c, k) for _ in range(3)) < 0.75 * serial for k in chosen):
            chosen.append(c)
    for c in cands:                                 # not enough independent queues: fill up
        if len(chosen) == n:
            break
        if c not in chosen:
            chosen.append(c)
    return chosen


class StaticBuffers:
    """Named device buffers with stable addresses; `stage` uploads new host data into them through a pinned host
    mirror with an asynchronous copy on the current stream.  (A pageable-memory copy blocks the host until everything
    queued before it on that stream has run, i.e. until the previous step has finished, which serialised host
    sampling with the device.)  The event per key bounds how far the host may run ahead: a mirror is not
    overwritten before the copy that read it has executed."""

    def __init__(self, device):
        self.device, self.bufs, self.generation = device, {}, 0
        self._pinned, self._events = {}, {}
        self.log = None         # a dict key -> [host array of every stage() call, in order] while a checker records the batches
                                # (tests / bench.py's loss parity: with the cross-iteration overlap the buffers of the
                                # discriminator steps already hold the NEXT iteration's batch when an iteration returns)
        self.replay = None      # a dict key -> [host arrays]: while set, the n-th stage() call of a key uploads the n-th array of its
                                # list INSTEAD of the caller's draw (tests: a single process re-runs the concatenated batches that
                                # the ranks of a data-parallel run logged); a key without entries left stages the caller's array

    def stage(self, key, array, dtype=None):
        if self.replay is not None and self.replay.get(key):
            array = self.replay[key].pop(0)
        t = torch.as_tensor(array)
        if dtype is not None:
            t = t.to(dtype)
        b = self.bufs.get(key)
        if b is None or b.shape != t.shape or b.dtype != t.dtype:
            b = torch.empty(t.shape, dtype=t.dtype, device=self.device)
            self.bufs[key] = b
            self._pinned[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._events[key] = None
            self.generation += 1                   # addresses changed: captured graphs are stale
        ev = self._events[key]
        if ev is not None:
            ev.synchronize()
        else:
            ev = self._events[key] = torch.cuda.Event()
        self._pinned[key].copy_(t)
        if self.log is not None:
            self.log.setdefault(key, []).append(self._pinned[key].numpy().copy())
        b.copy_(self._pinned[key], non_blocking=True)
        ev.record()
        return b

    def wait_staged(self, prefix):
        """Orders the CURRENT stream after the last upload of every key that starts with `prefix`: for a consumer that may run on
        another stream than the one the batch was staged on (batches staged ahead of their step on a sibling step's stream)."""
        cur = torch.cuda.current_stream()
        for key, ev in self._events.items():
            if ev is not None and key.startswith(prefix):
                cur.wait_event(ev)

    def __getitem__(self, key):
        return self.bufs[key]
