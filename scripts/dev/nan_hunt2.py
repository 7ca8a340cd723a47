"""Which gradient goes non-finite first?  Pipelined loop at the benchmark size, one sync + scan per iteration."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
model, real_set, synth_set, d_opt, g_opt, cfg = bench.setup(16, 256, 64)
model.use_graphs = True
model.overlap_discriminators = True
nets = {"generator": model.generator, "latent_regressor": model.latent_regressor, "synthetic_encoder": model.synthetic_encoder,
        "encoder": model.encoder, "discriminator": model.discriminator, "synth_discriminator": model.synth_discriminator,
        "latent_discriminator": model.latent_discriminator}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    out = model.training_iteration(real_set, synth_set, d_opt, g_opt)
    torch.cuda.synchronize()
    bad = False
    for name, net in nets.items():
        ga, wa = net.grad_arena, net.arena
        if not bool(torch.isfinite(ga).all()) or not bool(torch.isfinite(wa).all()):
            bad = True
            names = list(net.weight_names) if hasattr(net, "weight_names") else None
            for i, p in enumerate(net.weights):
                if p.grad is not None and not bool(torch.isfinite(p.grad).all()):
                    n_bad = int((~torch.isfinite(p.grad)).sum())
                    print("iteration", it, name, "grad", i, tuple(p.shape), "non-finite entries", n_bad, "of", p.numel(),
                          "nan", int(torch.isnan(p.grad).sum()), "inf", int(torch.isinf(p.grad).sum()), flush=True)
                if not bool(torch.isfinite(p).all()):
                    print("iteration", it, name, "WEIGHT", i, tuple(p.shape), "non-finite", int((~torch.isfinite(p)).sum()), flush=True)
    lossbad = [(s, k) for s, d in zip(("d", "sd", "ld", "g"), out) for k, v in d.items() if not np.isfinite(float(v))]
    if lossbad:
        print("iteration", it, "non-finite losses", lossbad[:6], flush=True)
    if bad or lossbad:
        gmax = {name: float(net.grad_arena[torch.isfinite(net.grad_arena)].abs().max()) for name, net in nets.items()}
        print("max finite |grad| per net", gmax)
        break
else:
    print("all finite")
