#!/usr/bin/env python
"""bench.py -- train-step images/sec at 256x256 (BASELINE.json metric).

One "step" = one whole reference training iteration of the SECOND stage (discriminator step +
synthetic-domain discriminator step + latent-discriminator step + generator step + EMA update,
reference confignet_second_stage.py:277-288) at 256x256, batch 16 per GPU, fp32, on synthetic
FFHQ-shaped uint8 pools that are resident in HBM before the timed region (BASELINE.json configs[1]).
images/sec = global batch / iteration time, the reference's own instrument
(confignet_first_stage.py:605-625).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, same per-GPU batch (weak scaling), gradient arenas all-reduced with RCCL.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak (~2.5 PF; the 5 PF figure is 2:1 sparsity)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # >= 20 timed iterations after >= 5 warm-up
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step function")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--pool", type=int, default=512, help="images in each synthetic uint8 pool")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="dispatch every kernel eagerly (no HIP graph replay)")
    ap.add_argument("--serial", action="store_true",
                    help="one stream, eager dispatch: no kernel overlaps another (the form to trace for per-kernel durations)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--no-parity", action="store_true", help="skip the loss-parity iteration (loss_parity_vs_cpu)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` block (bf16 loop, configs[3] / configs[4])")
    ap.add_argument("--timing-only", action="store_true",
                    help="the timed loop and nothing else (no per-function times, no roofline pass, no CPU baseline): the form the "
                         "`secondary` block runs the bf16 loop in")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: the reference's arithmetic (BASELINE.json configs[1], the headline).  bf16: configs[2]'s compute type "
                         "(bf16 activations / filter copies on bf16 MFMA, fp32 accumulate, fp32 master weights) -- a separate line")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N` (no launcher): start the N ranks ourselves, one process per GPU, exactly as
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...` would
        have = torch.cuda.device_count()
        if have < args.gpus and os.environ.get("CN_DP_SHARE_DEVICE", "0") != "1":
            sys.exit("bench.py: --gpus %d needs %d visible devices, this host has %d" % (args.gpus, args.gpus, have))
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--standalone",
               "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if env_world != args.gpus:
        sys.exit("bench.py: launched with WORLD_SIZE=%d but --gpus %d (the two must agree)" % (env_world, args.gpus))
    if torch.cuda.device_count() < int(os.environ.get("LOCAL_RANK", "0")) + 1 and os.environ.get("CN_DP_SHARE_DEVICE", "0") != "1":
        sys.exit("bench.py: rank with LOCAL_RANK=%s has no device (%d visible)" % (os.environ.get("LOCAL_RANK", "0"), torch.cuda.device_count()))
    from confignet_amd import ops, parallel

    ops.set_activation_dtype(args.dtype)
    world = parallel.init_from_env()
    rank = parallel.rank()
    if world > 1:                                     # N processes share the host: keep each rank's CPU thread pool small
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))
    if world == 1:
        torch.cuda.set_device(0)

    model, real_set, synth_set, d_opt, g_opt, cfg = setup(args.batch, args.res, args.pool, rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            parallel.barrier()                 # (an asynchronous RCCL work: nothing of it stays on a stream that will capture)
            torch.cuda.synchronize()

    # Every step's device half is one captured HIP graph (confignet_amd/graphs.py); with N > 1 the graph ends
    # after the backward pass and the RCCL all-reduce + Adam follow eagerly on the same stream.  Warm-up covers
    # the eager call + the capture of each graph.
    if args.serial:
        args.no_graphs = True
        model.fork_generator_step = False
    model.use_graphs = not args.no_graphs
    # real half of the next iteration's discriminator steps under the generator tail
    model.overlap_discriminators = model.use_graphs and os.environ.get("CN_NO_D_OVERLAP") is None
    # graph mode needs three untimed iterations whatever W says: eager call, capture, first replay (which still pays the
    # graph's one-off upload); the JSON reports the number actually run.  A capture that fails next to the process group is
    # FATAL (no silent eager fallback: an eager multi-GPU number would not be the configuration this benchmark names);
    # --no-graphs asks for eager dispatch explicitly and says so in config.dispatch.
    args.warmup = max(args.warmup, 3 if model.use_graphs else 0)
    for _ in range(args.warmup):
        model.training_iteration(real_set, synth_set, d_opt, g_opt)
    sync()
    if model.use_graphs:
        assert len(model._graphs) == 4 and all(g.graph is not None for g in model._graphs.values()), "step graphs were not captured"
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = model.training_iteration(real_set, synth_set, d_opt, g_opt)
    sync()
    elapsed = time.perf_counter() - t0
    finite = all(np.isfinite(float(l["loss_sum"].detach())) for l in losses)

    # Loss parity of THE DISPATCH JUST TIMED against the CPU oracle at this very size (rank 0, N = 1): one more iteration of the
    # same loop (replayed graphs, its real halves already in flight from the previous iteration) on fresh Adam moments; its
    # weights and batches go to the oracle subprocess of the cpu_baseline leg, which runs the same iteration first.
    if args.timing_only:
        if rank == 0:
            print(json.dumps({"value": round(args.batch * world / (elapsed / args.steps), 3), "unit": "images/sec", "dtype": args.dtype,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                              "losses_finite": bool(finite), "kernels_hash": kernels_hash()}), flush=True)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    parity_state = None
    if world == 1 and not args.no_cpu_baseline and not args.no_parity:
        parity_state = dump_parity_state(model, real_set, synth_set, d_opt, g_opt)

    # Per-step-function times: each function of the iteration on its own, in the dispatch mode
    # of the timed region, 5 calls each; "d_phase" is the three discriminator-type steps as the iteration runs them.
    def timed_ms(fn, reps=5):
        sync()
        t = time.perf_counter()
        for _ in range(reps):
            with model._main_line():
                fn()
        sync()
        return round(1e3 * (time.perf_counter() - t) / reps, 3)

    step_ms = {
        "discriminator": timed_ms(lambda: model.discriminator_training_step(real_set, d_opt)),
        "synth_discriminator": timed_ms(lambda: model.synth_discriminator_training_step(synth_set, d_opt)),
        "latent_discriminator": timed_ms(lambda: model.latent_discriminator_training_step(real_set, synth_set, d_opt)),
        "d_phase": timed_ms(lambda: model.run_concurrently([
            lambda: model.discriminator_training_step(real_set, d_opt),
            lambda: model.synth_discriminator_training_step(synth_set, d_opt),
            lambda: model.latent_discriminator_training_step(real_set, synth_set, d_opt)])),
        "generator": timed_ms(lambda: model.generator_training_step(real_set, synth_set, g_opt)),
    }

    # Roofline of the dominant kernel class, measured live with HIP events recorded on the launch stream
    # around every implicit-GEMM convolution launch of the SAME K iterations dispatched eagerly AND ON ONE STREAM
    # right after the timed region (event records cannot sit inside a replayed graph, and a per-kernel duration only
    # means something when the kernel has the GPU to itself; the kernels and shapes are identical).
    model.use_graphs = False
    model.fork_generator_step = False      # kernels of the two branches would overlap and stretch each other's events
    ops.prof_reset()
    ops.prof_enable(True)
    for _ in range(0 if os.environ.get("CN_BENCH_SKIP_ROOFLINE_PASS") == "1" else args.steps):   # (trace-only runs)
        model.training_iteration(real_set, synth_set, d_opt, g_opt)
    sync()
    ops.prof_enable(False)
    launches, kernel_ms, kernel_flops = ops.prof_collect()
    saved_flops = ops.prof_saved_flops()
    families = ops.prof_collect_by_family()
    nst = max(args.steps, 1)
    by_kernel = {k: {"launches_per_step": round(v["launches"] / nst, 1), "us_per_launch": round(1e3 * v["ms"] / v["launches"], 1),
                     "tflops": round(v["gflop"] / max(v["ms"], 1e-9), 2),
                     "algorithmic_mb_per_launch": round(v["algorithmic_mb"] / v["launches"], 2),
                     "algorithmic_gbps": round(v["algorithmic_mb"] / max(v["ms"], 1e-9), 1)}
                 for k, v in sorted(families.items(), key=lambda kv: -kv[1]["ms"])}
    alg_bytes_per_step = sum(v["algorithmic_mb"] for v in families.values()) * 1e6 / nst

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        parallel.all_reduce_max(t)
    elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.batch * world / (elapsed / args.steps)
        achieved = kernel_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        peak = FP32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
        out = {
            "metric": "train-step images/sec at 256x256 (G+D fwd+bwd)",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "ConfigNet second-stage iteration (D + synth-D + latent-D + G + EMA), %dx%d, "
                                   "batch %d per GPU, %s, Keras-Adam" % (args.res, args.res, args.batch,
                                                                          "fp32" if args.dtype == "f32" else "bf16 compute / fp32 master weights"),
                       "global_batch": args.batch * world, "resolution": args.res, "latent_dim": cfg_latent(model),
                       "parallelism": "dp%d" % world, "losses_finite": bool(finite),
                       "pipelining": ("steady state across iterations: the real half of the discriminator steps of iteration k+1 runs under "
                                      "the generator tail of iteration k, so the timed window holds the halves of iterations 2..K+1 instead "
                                      "of 1..K -- K iterations' worth of every kernel either way; CN_NO_D_OVERLAP=1 times the unpipelined loop"
                                      if model.overlap_discriminators else "none (every iteration starts after the previous one has finished)"),
                       "dispatch": ("eager, one stream" if args.serial else "eager" if args.no_graphs else
                                    "hip-graph replay per step function, D-type steps concurrent, G step forked over 2 streams" +
                                    (", real half of the next iteration's discriminator steps" + (" and the generator step's ground-truth VGG passes" if getattr(model, "_targets_ahead", None) is not None else "") + " under the generator tail" if model.overlap_discriminators else "") +
                                    ("" if not parallel.active() else " (fwd+bwd), eager %s all-reduce + Adam" %
                                     ("RCCL" if dist.get_backend() == "nccl" else "gloo (through the host: ranks share a device)")))},
            "step_functions_ms": step_ms,
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": pmc_traffic(args.dtype),
                         "mfma_busy_pmc": pmc_mfma_busy(args.dtype),
                         "kernel": ("fwd2/igemm_fwd/wgrad2/igemm_wgrad/wino_fwd/wino4_fwd (implicit-GEMM + Winograd F(2x2,3x3) / F(4x4,3x3) convolutions, v_mfma_f32_32x32x2_f32)" if args.dtype == "f32" else
                                    "fwd2<bf16>/igemm_bf16/igemm_bf16_wgrad (implicit-GEMM conv, v_mfma_f32_32x32x16_bf16) + the fp32 kernels of the "
                                    "3-channel image layers"),
                         "launches_per_step": launches / max(args.steps, 1),
                         "kernel_ms_per_step": round(kernel_ms / max(args.steps, 1), 3),
                         "measured": "HIP events around every launch of the class, same K iterations run serially (eager, one stream)",
                         "algorithmic_gflop_per_step": round(kernel_flops / max(args.steps, 1) / 1e9, 2),
                         "algorithmic_bytes_per_step": round(alg_bytes_per_step),
                         "hbm_frac_of_algorithmic_bytes": round(alg_bytes_per_step / max(kernel_ms / max(args.steps, 1) * 1e-3, 1e-12) / 8.0e12, 4),
                         "bound_note": ("fp32: the class is MFMA-bound by construction (%.2f TFLOP issued against %.0f GB of algorithmic bytes per "
                                        "step: %.0f FLOP/B, ridge 20 FLOP/B) and runs at the MFMA-busy fraction above; what separates it from the "
                                        "peak is per-launch overhead of 50-150 us kernels (ramp, tail, split-K) and operand delivery into LDS "
                                        "on the small tiles, see DESIGN.md section 3"
                                        % (kernel_flops / nst / 1e12, alg_bytes_per_step / 1e9, kernel_flops / nst / max(alg_bytes_per_step, 1.0))
                                        if args.dtype == "f32" else
                                        "bf16: neither roof governs -- 45 us average launches at 0.06 of the bf16 MFMA peak and 0.07 of the HBM peak "
                                        "on their algorithmic bytes: the class is bound by per-launch latency (ramp, tail, dependent chain), "
                                        "see DESIGN.md section 3"),
                         "algorithmic_bytes_per_launch": round(alg_bytes_per_step * max(args.steps, 1) / max(launches, 1)),
                         "traffic_over_algorithmic": (round(pmc_traffic(args.dtype) * launches / max(args.steps, 1) / alg_bytes_per_step, 2)
                                                      if (pmc_traffic(args.dtype) and alg_bytes_per_step) else None),
                         "by_kernel": by_kernel,
                         "work": "multiply-adds actually issued (Winograd: 16 per 2x2 tile resp. 36 per 4x4 tile, upsample-folded layers: the "
                                 "parity-class filters); `frac` is therefore comparable with mfma_busy_pmc -- an algorithm that issues "
                                 "fewer products for the same layer (F(4x4): 2.25 instead of 4 per output) LOWERS this figure while the "
                                 "iteration gets faster: see direct_equivalent",
                         "pipelined": {
                             "tflops": round(kernel_flops / nst / (ms_per_step * 1e-3) / 1e12, 3),
                             "frac": round(kernel_flops / nst / (ms_per_step * 1e-3) / 1e12 / peak, 4),
                             "note": "the same issued multiply-adds over the TIMED iteration (all lines concurrent, everything else of the "
                                     "iteration included): the matrix pipes' share of the wall clock.  `frac` above is per launch in "
                                     "ISOLATION -- since round 6 the Winograd kernels also take launches of 32 - 128 workgroups, which leave CUs "
                                     "idle when alone (lower `frac`) and are filled by the other lines in the pipelined loop (higher `value`)"},
                         "direct_equivalent": {
                             "gflop_per_step": round((kernel_flops + saved_flops) / max(args.steps, 1) / 1e9, 2),
                             "tflops": round((kernel_flops + saved_flops) / (kernel_ms * 1e-3) / 1e12, 3) if kernel_ms > 0 else 0.0,
                             "note": "the same launches priced as direct convolutions (round 1's definition of the algorithmic work, "
                                     "border taps of the Winograd layers counted): an algorithmic saving, NOT a roofline fraction"},
                         "recorded": "traffic / mfma_busy_pmc come from committed rocprofv3 PMC passes and are quoted only when "
                                     "profiles/round6_pmc_*.json carry this kernels_hash",
                         "kernels_hash": kernels_hash()},
            "torch_kernel_time_share": torch_kernel_share(),
            # kernel launches of one iteration, from the committed rocprofv3 trace of `bench.py --serial` (recorded, like the PMC fields)
            "launches_per_iteration": _recorded("round6_torch_share.json", "launches_per_iteration"),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref_losses = cpu_baseline(args, parity_state["path"] if parity_state else None)
            if parity_state is not None:
                out["loss_parity_vs_cpu"] = compare_losses(parity_state["losses"], ref_losses, parity_state["dispatch"])
                try:
                    os.remove(parity_state["path"])
                except OSError:
                    pass
        if world == 1 and not args.no_secondary and not args.no_cpu_baseline and args.dtype == "f32":     # (the full line only: auxiliary / traced runs pass --no-cpu-baseline)
            out["secondary"] = secondary(args)
        if not finite:
            # a timed loop whose losses went non-finite is not a measurement of the workload (round 4: such runs were also faster)
            out["invalid"] = "non-finite losses in the timed loop"
            print("bench.py: NON-FINITE LOSSES in the timed loop -- this line is not a valid measurement", file=sys.stderr, flush=True)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def setup(batch, res, pool, rank=0):
    """The benchmark's model, datasets and optimizers (BASELINE.json configs[1] at batch 16, res 256): seeded synthetic
    FFHQ-shaped uint8 pools resident in HBM, identical seeded weights on every rank."""
    from confignet_amd import ConfigNet, SyntheticFaceDataset, optim, parallel
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.confignet_utils import merge_configs
    np.random.seed(1234 + rank)                       # per-rank batch sampling stream
    real_set = SyntheticFaceDataset(pool, res, seed=1)
    synth_set = SyntheticFaceDataset(pool, res, seed=2)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": batch, "output_shape": (res, res, 3)})
    synth_set.process_metadata(cfg, True)
    cfg["image_loss_weight"] *= 10                    # second stage (train_confignet.py:67)
    model = ConfigNet(cfg, seed=0)                    # identical seeded weights on every rank
    parallel.broadcast_weights(model.all_networks())
    model.setup_training(None, synth_set, 0, real_training_set=real_set)
    d_opt, g_opt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
    model._pool(real_set), model._pool(synth_set)     # uint8 pools -> HBM before timing
    return model, real_set, synth_set, d_opt, g_opt, cfg


def kernels_hash():
    """sha1 over the HIP sources: recorded counter measurements are only quoted for the kernels they were taken on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "confignet_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def _recorded(name, key):
    """A counter-derived figure from the committed rocprofv3 PMC passes of this same command (hardware counters cannot be
    read from inside the benchmark process).  Returned only if the file was recorded on the CURRENT kernel sources
    (`kernels_hash`), else None: a stale measurement is not quoted."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fp:
            rec = json.load(fp)
        if rec.get("kernels_hash") != kernels_hash():
            return None
        return rec[key]
    except Exception:
        return None


def pmc_traffic(dtype="f32"):
    """HBM bytes per launch of the dominant kernel class (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE;
    scripts/pmc_summary.py, scripts/pmc_traffic_json.py)."""
    v = _recorded("round6_pmc_traffic%s.json" % ("" if dtype == "f32" else "_" + dtype), "hbm_bytes_per_launch")
    return None if v is None else round(v)


def pmc_mfma_busy(dtype="f32"):
    """MFMA-pipe busy fraction of the class (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE; scripts/pmc_mfma.py)."""
    v = _recorded("round6_pmc_mfma%s.json" % ("" if dtype == "f32" else "_" + dtype), "mfma_busy_fraction")
    return None if v is None else round(v, 4)


def torch_kernel_share():
    """Share of the GPU time of one iteration spent in PyTorch's own kernels (autograd's gradient accumulation adds, cat,
    fills, small (N, L) algebra) from the committed kernel trace of this command -- north_star: torch is plumbing."""
    v = _recorded("round6_torch_share.json", "torch_kernel_time_share")
    return None if v is None else round(v, 4)


def cpu_baseline(args, state_path=None):
    """The oracle's restatement of one whole second-stage iteration, timed on the host cores in a subprocess with a hard
    time limit: 1 warm-up iteration on 16 threads (it pays the one-off costs), then ONE iteration AT THE BENCHMARK'S BATCH on each of
    {16, 32, 64, 128} threads (pinned: OMP_PLACES=cores, OMP_PROC_BIND=close; 25 s limit each), then the median of 3 iterations at the
    fastest count.  state_path: a dump_parity_state() file -- the warm-up iteration then runs on the HIP path's own weights and batch
    and its loss dicts come back as the second return value."""
    import subprocess
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", str(args.cpu_batch), str(args.res)] + ([state_path] if state_path else [])
    try:
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, OMP_PLACES="cores", OMP_PROC_BIND="close"))
        lines = p.stdout.strip().splitlines()
        if not lines:                  # the subprocess died: say how, with the end of what it wrote to stderr
            raise RuntimeError("oracle.cpu_baseline exited with code %s and no result; stderr ends: %s"
                               % (p.returncode, " | ".join(p.stderr.strip().splitlines()[-6:])[-900:]))
        r = json.loads(lines[-1])
        return {"value": round(r["value"], 4), "unit": "images/sec", "cores": r["cores"], "kind": "port",
                "sample": "second-stage iteration at %dx%d, batch %d: 1 warm-up + median of 3 (%.1f s each) on %d threads of the %d host cores, "
                          "pinned (OMP_PLACES=cores, OMP_PROC_BIND=close); thread count = the fastest of one iteration each at this batch on "
                          "{16, 32, 64, 128} threads after the warm-up (25 s limit per probe): %s s; torch-CPU fp32 restatement of the reference (oracle/) -- TensorFlow 2.1 itself "
                          "cannot be installed here"
                          % (args.res, args.res, args.cpu_batch, r["seconds"], r["cores"], r["host_cores"], r["thread_probe_seconds"])}, \
            r.get("parity_losses")
    except Exception as e:   # timeout / crash: report it, never block the GPU result
        return {"value": None, "unit": "images/sec", "cores": None, "kind": "port", "sample": "failed: %r" % (e,)}, None


def secondary(args):
    """The other BASELINE.json configurations, measured by THIS command right after the headline (the GPU is idle again; each runs
    in a child process because the activation dtype is process-wide state): configs[2]'s compute type -- the same loop, same
    steps / warm-up, in bf16 -- and scripts/bench_configs.py: configs[3] (one-shot fine-tune, 256x256, 200 steps: the reference's
    literal loop and the cached-target form), configs[4] (LatentGAN at batch 4096: host sampling as the reference, device
    sampling) and the first-stage iteration at 128x128 batch 8.  Never blocks the headline: a failure is reported in place."""
    import subprocess

    def child(cmd, limit):
        try:
            p = subprocess.run([sys.executable] + cmd, cwd=ROOT, capture_output=True, text=True, timeout=limit)
            lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
            if not lines:
                return {"failed": "exit code %s; stderr ends: %s" % (p.returncode, " | ".join(p.stderr.strip().splitlines()[-4:])[-600:])}
            return json.loads(lines[-1])
        except Exception as e:
            return {"failed": repr(e)}

    t0 = time.perf_counter()
    out = {"bf16": child([os.path.abspath(__file__), "--dtype", "bf16", "--steps", str(args.steps), "--warmup", str(args.warmup),
                          "--batch", str(args.batch), "--res", str(args.res), "--pool", str(args.pool), "--timing-only"], 300),
           "configs": child([os.path.join(ROOT, "scripts", "bench_configs.py")], 300)}
    out["bf16"]["what"] = ("BASELINE.json configs[2]'s compute type on one GPU: the headline's loop with bf16 activations / filter copies on "
                           "v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 master weights; a separate figure, never the headline")
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


PARITY_NETS = ("generator", "generator_smoothed", "discriminator", "synth_discriminator", "latent_discriminator",
               "latent_regressor", "synthetic_encoder", "real_encoder")


def dump_parity_state(model, real_set, synth_set, d_opt, g_opt, path=None):
    """Runs ONE MORE iteration in the dispatch the caller has set up and writes what the CPU oracle needs to run the same
    iteration (oracle/cpu_baseline.py:load_state): every network's weights before it, the batches (uint8 images gathered ON
    THE HOST from the dataset arrays by the staged indices, flip flags, face-model parameters, rotations), and the three
    discriminators' weights after it.  Adam moments / step counters are reset first (the oracle starts from fresh moments);
    weights are NOT touched, so with the cross-iteration overlap the real halves that the previous iteration pre-replayed for
    this one are consumed as they are -- this is the pipelined dispatch, not a restart.  Returns path / losses / dispatch."""
    import tempfile

    import torch
    nets = dict(zip(PARITY_NETS, (model.generator, model.generator_smoothed, model.discriminator, model.synth_discriminator,
                                  model.latent_discriminator, model.latent_regressor, model.synthetic_encoder, model.encoder)))
    model._bufs.log = {}
    model.training_iteration(real_set, synth_set, d_opt, g_opt)           # (its tail stages the discriminator batches of the next one)
    torch.cuda.synchronize()
    pipelined = bool(model.use_graphs and model.overlap_discriminators and
                     all(getattr(g, "prelaunched", False) for g in model._graphs.values() if g.name in ("d", "sd")))
    # (the latent-discriminator and generator steps' host halves move with them when the generator step's ground-truth VGG
    # passes are replayed ahead: ConfigNet._prelaunch_generator_targets)
    ahead = (("d/", "sd/") if pipelined else ()) + (("ld/", "g/") if (pipelined and "g" in model._prestaged) else ())
    for o in (d_opt, g_opt):
        o.iterations = 0
        for mom, var in o._state.values():
            mom.zero_()
            var.zero_()
    z = {}
    for name, net in nets.items():
        ws = net.get_weights()
        z["n/" + name] = np.int64(len(ws))
        for i, w in enumerate(ws):
            z["%s/%d" % (name, i)] = w
    vgg = model.perceptual_loss._pretrained_dnn_activations.get_weights()
    z["n/vgg"] = np.int64(len(vgg))
    for i, w in enumerate(vgg):
        z["vgg/%d" % i] = w
    mark = {k: len(v) for k, v in model._bufs.log.items()}
    out = model.training_iteration(real_set, synth_set, d_opt, g_opt)
    losses = [{k: float(v) for k, v in d.items()} for d in out]
    log, model._bufs.log = model._bufs.log, None
    # this iteration's batch per key: what it staged itself (first entry after `mark`), except the image-discriminator steps
    # of the pipelined loop, whose batch was staged by the PREVIOUS iteration's tail (last entry before `mark`)
    B = {k: v[mark[k] - 1 if (ahead and k.startswith(ahead)) else mark[k]] for k, v in log.items()}
    assert len(log["g/rot"]) == mark["g/rot"] + 1 and len(log["d/real_idx"]) >= mark["d/real_idx"] + 1, "unexpected staging sequence"
    for name in ("discriminator", "synth_discriminator", "latent_discriminator"):
        for i, w in enumerate(nets[name].get_weights()):
            z["post/%s/%d" % (name, i)] = w
    R, Sy = np.asarray(real_set.imgs), np.asarray(synth_set.imgs)
    z.update({"img/real_d": R[B["d/real_idx"]], "flip/real_d": B["d/real_flip"], "img/enc_in_d": R[B["d/enc_idx"]],
              "img/real_sd": Sy[B["sd/real_idx"]], "flip/real_sd": B["sd/real_flip"], "sd/rot": B["sd/rot"],
              "img/real_ld": R[B["ld/real_idx"]], "flip/real_ld": B["ld/real_flip"],
              "g/rot": B["g/rot"], "img/synth_g": Sy[B["g/synth_idx"]], "eye_masks_g": np.asarray(synth_set.eye_masks)[B["g/synth_idx"]],
              "img/real_g": R[B["g/real_idx"]], "flip/real_g": B["g/real_flip"]})
    for k, v in B.items():
        if "/p/" in k:
            z[k] = v
    if path is None:
        fd, path = tempfile.mkstemp(prefix="cn_parity_", suffix=".npz")
        os.close(fd)
    np.savez(path, **z)
    dispatch = ("hip-graph replay, real halves of this iteration's discriminator steps" + (" and the generator step's ground-truth VGG passes" if "g/" in ahead else "") + " pre-replayed under the previous generator tail"
                if pipelined else "hip-graph replay" if model.use_graphs else "eager")
    return {"path": path, "losses": losses, "dispatch": dispatch}


def compare_losses(got, ref, dispatch, tol=1e-3):
    """max over every scalar of the four loss dicts of |hip - cpu| / max(1, |cpu|) (north_star: 1e-3 at fp32)."""
    if ref is None:
        return {"max_err": None, "tolerance": tol, "note": "the CPU oracle run failed"}
    import math
    worst, n, bad = (0.0, None), 0, []
    for g, step in zip(got, ("d", "synth_d", "latent_d", "g")):
        r = ref[step]
        assert list(g.keys()) == list(r.keys()), (step, list(g.keys()), list(r.keys()))
        for k, v in g.items():
            n += 1
            e = abs(v - r[k]) / max(1.0, abs(r[k]))
            if not math.isfinite(e):               # a NaN / inf scalar on either side fails the gate whatever else was seen
                bad.append("%s/%s: hip %r cpu %r" % (step, k, v, r[k]))
            elif e > worst[0]:
                worst = (e, "%s/%s: hip %.6g cpu %.6g" % (step, k, v, r[k]))
    return {"max_err": float("%.3g" % worst[0]) if not bad else None, "tolerance": tol, "ok": bool(not bad and worst[0] <= tol),
            "n_scalars": n, "worst": worst[1] if not bad else None, "non_finite": bad,
            "what": "all loss scalars of one whole iteration at the benchmark's size, HIP (%s) vs the torch-CPU fp32 oracle on the "
                    "same weights and batches; |hip - cpu| / max(1, |cpu|)" % dispatch}


def cfg_latent(model):
    return int(model.config["latent_dim"])


if __name__ == "__main__":
    main()
