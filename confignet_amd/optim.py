"""Keras-form Adam on flat parameter arenas ([TF-2.1] R10):
theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps), eps = 1e-7, with ONE iteration counter per
optimizer object shared by every network it updates (confignet_first_stage.py:601-610: the same
discriminator_optimizer serves D, synth-D and latent-D in turn, so t advances 3x per iteration)."""
import math

import torch

from . import ops
from . import parallel


_deferred = None     # list of (optimizer, nets, slot) while a step is recorded with deferred updates
_touched = None      # list that receives every network an Adam launch updates (graphs.StepGraph sets it while it captures a step)


class deferred_updates:
    """Inside this context apply_gradients() only RECORDS its (optimizer, nets, slot): used when a step's device
    half is captured into a HIP graph for a multi-rank run, where the gradient all-reduce (RCCL, not capturable
    together with the compute here) and the Adam launches that depend on it are issued eagerly after each replay
    (`run_deferred`)."""

    def __enter__(self):
        global _deferred
        self.prev, _deferred = _deferred, []
        self.items = _deferred
        return self.items

    def __exit__(self, *exc):
        global _deferred
        _deferred = self.prev


def run_deferred(items):
    for opt, nets, slot in items:
        opt.apply_gradients(nets, advance=False, slot=slot)


class Adam:
    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, **_):
        assert not amsgrad, "amsgrad is False everywhere in the reference"
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon
        self.iterations = 0
        self._state = {}
        self._checked_lists = {}           # network -> {variable list: epoch of the network's moments it was checked at}
        self._ever_listed = {}             # network -> ids of the weights any Keras-form call has listed
        self._list_epoch = {}              # network -> bumped whenever a call lists a weight for the first time
        self._lr_dev = {}

    def lr_t(self):
        t = self.iterations
        return self.lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def advance(self, slot="default"):
        """Host half of one Keras apply_gradients call: t += 1 and the new lr_t is written to a device scalar.
        Kept apart from apply_gradients so that the device half can live inside a captured HIP graph.  Each
        step function owns a `slot` (its own scalar), so several captured steps that share this optimizer can
        be in flight at once, each with the lr_t of ITS position in the shared step count."""
        if slot not in self._lr_dev:
            self._lr_dev[slot] = torch.zeros(1, device="cuda" if torch.cuda.is_available() else "cpu", dtype=torch.float32)
        self.iterations += 1
        self._lr_dev[slot].fill_(self.lr_t())

    def state_for(self, net):
        st = self._state.get(id(net))
        if st is None:
            st = (torch.zeros_like(net.arena), torch.zeros_like(net.arena))
            self._state[id(net)] = st
        return st

    def apply_gradients(self, nets, advance=True, slot="default"):
        """One Keras apply_gradients call over the flat arenas of `nets` (a Net or a list of Nets):
        data-parallel gradient all-reduce first, then one fused Adam launch per arena.
        Also accepts the Keras form `apply_gradients(zip(gradients, variables))` (confignet_first_stage.py:472-474) when the
        variables are weights of confignet_amd networks: the gradients are copied into the owners' gradient arenas (weights
        of those networks that are not listed get a zero gradient, which Keras-Adam with zero moments leaves unchanged).
        RESTRICTION of that form: Adam runs over the owners' WHOLE arenas, so an unlisted weight of a listed network whose
        moments are already non-zero in THIS optimizer (it was listed in an earlier call) would keep moving, which Keras does
        not do -- asserted below on the second-moment arena, once per variable list (the reference never mixes variable lists on one
        optimizer)."""
        if not isinstance(nets, (list, tuple)) and hasattr(nets, "__iter__") and not hasattr(nets, "arena"):
            nets = list(nets)                   # zip(...) and other iterators
        if isinstance(nets, (list, tuple)) and nets and isinstance(nets[0], (list, tuple)):
            pairs, nets = nets, []
            for _, var in pairs:
                owner = getattr(var, "_cn_owner", None)
                assert owner is not None, "apply_gradients(zip(grads, vars)): variables must be weights of a confignet_amd network"
                if all(owner is not n for n in nets):
                    nets.append(owner)
                    owner.grad_arena.zero_()
            for g, var in pairs:
                if g is not None:
                    var.grad.copy_(g.reshape(var.shape))
            listed = frozenset(id(var) for _, var in pairs)
            for owner in nets:
                st = self._state.get(id(owner))
                # A list is checked once per STATE of the owner's moments.  The state changes only when a call lists a weight that
                # no earlier call had listed (new moments appear): that bumps the owner's epoch and every list checked before it is
                # checked again on its next use -- A, superset B, A again still fails on the third call.  Calls that repeat or
                # alternate already-seen lists do NOT re-run the masked reduction (a host synchronisation, illegal inside a
                # HIP-graph capture).
                oid = id(owner)
                ever = self._ever_listed.setdefault(oid, set())
                if not listed <= ever:
                    ever |= listed
                    self._list_epoch[oid] = self._list_epoch.get(oid, 0) + 1
                seen = self._checked_lists.setdefault(oid, {})
                if st is None or seen.get(listed) == self._list_epoch[oid]:
                    continue
                # once per (optimizer, variable list): one masked reduction over the SECOND-moment arena (v > 0 wherever a
                # weight has ever had a non-zero gradient in this optimizer) restricted to the unlisted weights
                mask = torch.zeros_like(owner.arena, dtype=torch.bool)
                for w in owner.trainable_weights:
                    if id(w) not in listed and w.numel():
                        off = (w.data_ptr() - owner.arena.data_ptr()) // 4
                        mask[off:off + w.numel()] = True
                assert not bool((st[1].ne(0) & mask).any()), \
                    "apply_gradients(zip(grads, vars)): an unlisted weight of a listed network has optimizer state"
                seen[listed] = self._list_epoch[oid]
        if not isinstance(nets, (list, tuple)):
            nets = [nets]
        if _deferred is not None:
            assert not advance
            _deferred.append((self, list(nets), slot))
            return
        if advance:
            self.advance(slot)
        parallel.allreduce_gradients(nets)
        for net in nets:
            m, v = self.state_for(net)
            ops.adam_step(net.arena, net.grad_arena, m, v, None, self._lr_dev[slot], self.beta_1, self.beta_2, self.epsilon)
            net.mark_updated()
            if _touched is not None and all(net is not t for t in _touched):
                _touched.append(net)
