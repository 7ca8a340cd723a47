"""Network container: Keras-ordered weight lists stored in one flat HBM arena per network.

Mirrors the protocol the reference's callers use on keras Models: `net(inputs)`,
`.predict(ndarray)`, `.get_weights()`, `.set_weights(list)`, `.trainable_weights`.  All
trainable tensors of a network are views into ONE contiguous fp32 arena with a matching
gradient arena, so the optimizer is one kernel launch and the data-parallel gradient exchange
is one RCCL all-reduce per network."""
import math

import numpy as np
import torch


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("confignet_amd needs an MI355X GPU: the hot path is HIP-only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def glorot_uniform(rng, shape):
    """Keras default kernel initializer (R4)."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = rf * shape[-2], rf * shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def he_normal(rng, shape):
    """Stand-in init for the pretrained keras.applications stacks (no imagenet weights offline):
    keeps activation magnitudes stable through 10-50 ReLU layers."""
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    return (rng.standard_normal(size=shape) * math.sqrt(2.0 / (rf * shape[-2]))).astype(np.float32)


WEIGHTS_EPOCH = [0]     # global generation of derived weight caches: bumped when a HIP-graph capture starts / ends (copies made
                        # inside a capture live in that graph's pool)


class Net:
    epoch = 0           # bumped whenever THIS network's weights change in place (optimizer, set_weights, EMA, broadcast)

    def mark_updated(self):
        """Raw-pointer kernels (Adam, EMA) do not bump torch's version counters: derived caches (the tap-flipped fp32 filter
        of the data-gradient GEMM, the bf16 operand copies) are keyed on this per-network epoch instead, so an update of one
        network does not invalidate the copies of the others (the frozen VGG stacks keep theirs for the whole run)."""
        self.epoch += 1

    def non_trainable_changed(self):
        """Hook: the non-trainable tensors (BatchNorm moving statistics) were rewritten in place."""

    def __init__(self):
        self._entries = []          # (name, np array, trainable)
        self.weights = []           # torch tensors, Keras get_weights() order
        self._trainable_idx = []
        self.arena = None           # flat parameters
        self.grad_arena = None
        self.device = None

    # ---- construction -----------------------------------------------------------------------
    def add_weight(self, name, array, trainable=True):
        self._entries.append((name, np.ascontiguousarray(array, dtype=np.float32), trainable))
        return len(self._entries) - 1

    def finalize(self):
        self.device = require_gpu()
        n_train = sum(e[1].size for e in self._entries if e[2])
        # 16-byte aligned slots so every view supports float4 access
        offs, cur = [], 0
        for _, a, tr in self._entries:
            if tr:
                offs.append(cur)
                cur += (a.size + 3) // 4 * 4
            else:
                offs.append(-1)
        self.arena = torch.zeros(max(cur, 4), device=self.device, dtype=torch.float32)
        self.grad_arena = torch.zeros_like(self.arena)
        self.weights, self._trainable_idx = [], []
        for i, ((name, a, tr), off) in enumerate(zip(self._entries, offs)):
            if tr:
                p = self.arena[off:off + a.size].view(a.shape)
                p.copy_(torch.from_numpy(a))
                p.requires_grad_(True)
                p.grad = self.grad_arena[off:off + a.size].view(a.shape)
                self._trainable_idx.append(i)
            else:
                p = torch.from_numpy(a).to(self.device)
            p._cn_owner = self
            self.weights.append(p)
        self.n_trainable = n_train
        return self

    # ---- keras-like protocol ----------------------------------------------------------------
    @property
    def trainable_weights(self):
        return [self.weights[i] for i in self._trainable_idx]

    def get_weights(self):
        return [w.detach().cpu().numpy().copy() for w in self.weights]

    def set_weights(self, weights):
        weights = list(weights)
        assert len(weights) == len(self.weights), "expected %d arrays, got %d" % (len(self.weights), len(weights))
        with torch.no_grad():
            for w, a in zip(self.weights, weights):
                a = np.asarray(a, dtype=np.float32)
                assert tuple(a.shape) == tuple(w.shape), "shape mismatch %s vs %s" % (a.shape, tuple(w.shape))
                w.copy_(torch.from_numpy(np.ascontiguousarray(a)))
        self.mark_updated()
        self.non_trainable_changed()

    def copy_weights_from(self, other):
        with torch.no_grad():
            self.arena.copy_(other.arena)
            for w, o in zip(self.weights, other.weights):
                if not w.requires_grad:
                    w.copy_(o)
        self.mark_updated()
        self.non_trainable_changed()

    def zero_grad(self):
        self.grad_arena.zero_()

    def requires_grad_(self, flag):
        for i in self._trainable_idx:
            self.weights[i].requires_grad_(flag)
        return self

    def to_device(self, x, dtype=torch.float32):
        if torch.is_tensor(x):
            return x.to(device=self.device, dtype=dtype)
        return torch.as_tensor(np.ascontiguousarray(x)).to(device=self.device, dtype=dtype)

    def predict(self, x, batch_size=32):
        """keras Model.predict: batches of 32, numpy out (R11)."""
        raise NotImplementedError


def backward_into_arenas(loss, nets, extra=(), grad_outputs=None, accumulate=False):
    """tape.gradient(loss, trainable_weights) written into the networks' gradient arenas.
    Uses autograd.grad + one multi-tensor copy instead of .backward(): AccumulateGrad nodes are bound to the
    stream they were created on, which breaks HIP-graph capture of a step on a capture stream.
    Round 3: the arenas are cleared by one launch each and every gradient is ADDED -- the convolution / dense weight and bias
    gradients by their own kernels, on a side stream off the backward chain (ops.grad_sink), the rest (norm parameters, free
    variables) by one multi-tensor add of what autograd returns.
    Weights switched off with requires_grad_(False) (a variable the caller leaves out of the reference's
    trainable list, e.g. the expression slice of fine_tune_on_img(force_neutral_expression=True)) get a zero
    gradient: Keras-Adam on a zero gradient with zero moments leaves them unchanged.
    `extra`: further tensors of the tape whose gradients are returned (a list, None where unused) -- with `grad_outputs` and
    `loss` a list of such tensors a later call continues the backward pass from them (two-part backward, see
    ConfigNetFirstStage._generator_update).  accumulate: ADD to the arenas (second part of a loss whose first part was
    written by an earlier call: the discriminator steps' real / fake halves)."""
    from . import ops
    every = [p for n in nets for p in n.trainable_weights]
    params = [p for p in every if p.requires_grad]
    extra = list(extra)
    if not accumulate:
        for n in nets:                      # everything below ADDS into the arenas (kernels with an accumulate mode, then one
            ops.zero_(n.grad_arena)         # multi-tensor add for what autograd returns)
    with ops.grad_sink(params):             # filter / dense / bias gradients go straight into their arena slots (ops.grad_sink)
        grads = torch.autograd.grad(loss, params + extra, grad_outputs=grad_outputs, allow_unused=True)
    extra_grads = list(grads[len(params):])
    grads = grads[:len(params)]
    dst = [p.grad for p, g in zip(params, grads) if g is not None]
    src = [g.reshape(p.shape) for p, g in zip(params, grads) if g is not None]
    if dst:
        torch._foreach_add_(dst, src)
    return extra_grads
