"""Helper of test_bf16_gpu.py::test_bf16_data_parallel_dispatch_at_full_size_matches_cpu_oracle: BASELINE.json configs[2]'s
per-rank workload -- the second-stage iteration at 256x256, batch 16 per rank, bf16 compute -- in the benchmark's dispatch
(bench.setup, step graphs + cross-iteration overlap), optionally on a 1-rank RCCL group (CN_FORCE_DP=1: graphs that end after
the backward pass, eager all-reduce + Adam, the two-part generator backward) and with the global batch statistics.  Writes the
loss scalars of one whole iteration, the fp32 CPU oracle's scalars for the same weights and batches, and the discriminators'
weights after the iteration."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, global_stats, dtype):
    from confignet_amd import ops, parallel
    import bench
    ops.set_activation_dtype(dtype)
    parallel.init_from_env()
    model, real_set, synth_set, d_opt, g_opt, _ = bench.setup(16, 256, 64)
    model.config["dp_global_batch_statistics"] = bool(global_stats)
    model.use_graphs = True
    model.overlap_discriminators = True
    for _ in range(4):                                         # eager warm-ups, capture, first concurrent replay (+ prestage)
        model.training_iteration(real_set, synth_set, d_opt, g_opt)
    torch.cuda.synchronize()
    assert all(g.graph is not None for g in model._graphs.values())
    st = bench.dump_parity_state(model, real_set, synth_set, d_opt, g_opt)
    try:
        p = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "16", "256", st["path"], "--parity-only"],
                           cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        ref = json.loads(p.stdout.strip().splitlines()[-1])["parity_losses"]
        z = np.load(st["path"])
        post = {k.replace("/", "_"): z[k] for k in z.files if k.startswith("post/")}
    finally:
        os.remove(st["path"])
    with open(out_path + ".json", "w") as fp:
        json.dump({"losses": st["losses"], "ref": ref, "dispatch": st["dispatch"], "dp": bool(parallel.active()),
                   "split": [bool(g.split) for g in model._graphs.values()]}, fp)
    np.savez(out_path, **post)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "bf16")
