"""(scripts/step_ops.py: g_step_ops.py for any of the four steps: argv[1] in D / SD / LD / G.)
Every launch-producing call of ONE generator step (second stage, 256x256, batch 16), eager, in order of frequency: library entry
points (ctypes) and torch's own operators with their shapes -- the list behind the launch-count work of DESIGN.md section 3.
    python scripts/g_step_ops.py"""
import collections
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from confignet_amd import _lib

model, real_set, synth_set, d_opt, g_opt, cfg = bench.setup(16, 256, 64)
model.use_graphs = False
for _ in range(2):
    model.training_iteration(real_set, synth_set, d_opt, g_opt)
torch.cuda.synchronize()

STEP = sys.argv[1] if len(sys.argv) > 1 else "G"
RUN = {"D": lambda: model.discriminator_training_step(real_set, d_opt), "SD": lambda: model.synth_discriminator_training_step(synth_set, d_opt),
       "LD": lambda: model.latent_discriminator_training_step(real_set, synth_set, d_opt), "G": lambda: model.generator_training_step(real_set, synth_set, g_opt)}[STEP]
lib_calls = collections.Counter()
for name in _lib.SIGNATURES:
    fn = getattr(_lib.lib, name)

    def wrap(*a, _fn=fn, _n=name):
        lib_calls[_n] += 1
        return _fn(*a)
    setattr(_lib.lib, name, wrap)
import confignet_amd.ops as ops
ops.lib = _lib.lib

torch_calls = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        shapes = tuple(tuple(a.shape) for a in args if torch.is_tensor(a))[:3]
        dev = any(torch.is_tensor(a) and a.is_cuda for a in args) or (torch.is_tensor(out) and out.is_cuda)
        if dev:
            torch_calls[(str(func), shapes)] += 1
        return out


with Log():
    with model._main_line():
        RUN()
torch.cuda.synchronize()
# every thread's operators (the backward pass runs on autograd's device thread, which the dispatch mode above does not see)
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    with model._main_line():
        RUN()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
skip = ("view", "reshape", "detach", "alias", "expand", "transpose", "slice", "select", "unsqueeze", "squeeze", "permute", "as_strided",
        "split", "unbind", "_unsafe_view", "empty", "record_stream", "set_", "is_", "to", "item", "_local_scalar", "resolve", "lift", "contiguous", "t", "narrow", "chunk", "flatten", "result_type", "numpy")
rows = [e for e in rows if e.key[6:] not in skip and not e.key[6:].startswith(("empty", "_reshape", "view"))]
print("torch operators of one %s step" % STEP + ", all threads (profiler): %d calls" % sum(e.count for e in rows))
for e in sorted(rows, key=lambda e: -e.count)[:70]:
    print("%5d  %-28s %s" % (e.count, e.key, str(e.input_shapes)[:110]))
print("library entry points: %d calls" % sum(lib_calls.values()))
for k, v in lib_calls.most_common(40):
    print("%5d  %s" % (v, k))
view_like = ("view", "reshape", "detach", "alias", "expand", "t.default", "transpose", "slice", "select", "unsqueeze", "squeeze", "permute", "as_strided", "split", "unbind", "_unsafe_view", "is_", "size", "stride", "empty", "record_stream", "set_")
dev_calls = {k: v for k, v in torch_calls.items() if not any(t in k[0] for t in view_like)}
print("torch operators that launch (views / allocations left out): %d calls" % sum(dev_calls.values()))
agg = collections.Counter()
for (f, sh), v in dev_calls.items():
    agg[f] += v
for k, v in agg.most_common(25):
    print("%5d  %s" % (v, k))
print("by shape:")
for (f, sh), v in sorted(dev_calls.items(), key=lambda kv: -kv[1])[:60]:
    print("%5d  %-40s %s" % (v, f, sh))
