"""The generator step of the second-stage iteration (256x256, batch 16) ALONE, dispatched eagerly on one stream: the command to
put under `rocprofv3 --kernel-trace --stats` for the kernel composition of the iteration's backbone (scripts/prof_summary.py).
    python scripts/g_step_trace.py [reps]"""
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
model, real_set, synth_set, d_opt, g_opt, cfg = bench.setup(16, 256, 64)
model.use_graphs = False
model.fork_generator_step = False
for _ in range(2):
    model.training_iteration(real_set, synth_set, d_opt, g_opt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    with model._main_line():
        model.generator_training_step(real_set, synth_set, g_opt)
e1.record()
torch.cuda.synchronize()
print("generator step, eager, one stream: %.2f ms per call (host-bound dispatch; kernel time is what the trace holds)" % (e0.elapsed_time(e1) / reps))
