"""Secondary BASELINE.json configs (parity-test cases, not bench lines): config 4 = one-shot generator fine-tune
loop (256x256, 1 image, 200 steps), config 5 = LatentGAN D+G step at batch 4096; config 0' = first-stage iteration at
128x128 batch 8 (the reference generator cannot emit 64x64)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from confignet_amd import ConfigNet, ConfigNetFirstStage, LatentGAN, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs
out = {}
np.random.seed(0)
# config 4
ds = SyntheticFaceDataset(8, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 2, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
m.use_graphs = True
m.fine_tune_on_img(ds.imgs[0], n_iters=3)
torch.cuda.synchronize(); t0 = time.perf_counter()
emb, rot = m.fine_tune_on_img(ds.imgs[0], n_iters=200)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["config4_finetune_256_1img_200steps"] = {
    "seconds": round(dt, 3), "steps_per_sec": round(200 / dt, 2), "finite": bool(np.isfinite(emb).all()),
    "target_features": "cached",
    "note": "ALGORITHMIC SAVING, not kernel speed: the target image's VGG-19 / VGGFace activations are computed once before the loop "
            "(identical values; the reference recomputes them every step, confignet_second_stage.py:369-370) -- 2 of the 6 VGG passes "
            "per step are skipped.  The literal loop is the next entry."}
m.cache_target_features = False                      # the reference's literal loop: target features recomputed every step
m.fine_tune_on_img(ds.imgs[0], n_iters=3)
torch.cuda.synchronize(); t0 = time.perf_counter()
emb, rot = m.fine_tune_on_img(ds.imgs[0], n_iters=200)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["config4_finetune_256_1img_200steps_literal"] = {"seconds": round(dt, 3), "steps_per_sec": round(200 / dt, 2), "finite": bool(np.isfinite(emb).all()),
                                                      "target_features": "recomputed every step (as the reference does)"}
del m
# config 5
gan = LatentGAN({"latent_dim": 145, "batch_size": 4096}, seed=0)
opt = optim.Adam(**gan.config["optimizer"])
emb = np.random.normal(size=(20000, 145)).astype(np.float32)
for _ in range(5):
    gan.discriminator_training_step(emb, opt); gan.generator_training_step(opt); gan.update_smoothed_weights()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    d = gan.discriminator_training_step(emb, opt); g = gan.generator_training_step(opt); gan.update_smoothed_weights()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["config5_latentgan_b4096"] = {"steps_per_sec": round(50 / dt, 2), "ms_per_step": round(1e3 * dt / 50, 3), "finite": bool(np.isfinite(float(d["loss_sum"]))),
                                   "note": "np.random.normal of the (4096,145) latents on the host (as the reference does) is ~29 ms per draw, 2 draws per step"}
gan = LatentGAN({"latent_dim": 145, "batch_size": 4096, "device_latent_sampling": True}, seed=0)
opt = optim.Adam(**gan.config["optimizer"])
emb_dev = torch.as_tensor(emb).cuda()
for _ in range(5):
    gan.discriminator_training_step(emb_dev, opt); gan.generator_training_step(opt); gan.update_smoothed_weights()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    d = gan.discriminator_training_step(emb_dev, opt); g = gan.generator_training_step(opt); gan.update_smoothed_weights()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["config5_latentgan_b4096_device_sampling"] = {"steps_per_sec": round(50 / dt, 2), "ms_per_step": round(1e3 * dt / 50, 3), "finite": bool(np.isfinite(float(d["loss_sum"])))}
# config 0' first-stage 128, batch 8
ds = SyntheticFaceDataset(64, 128, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 8, "output_shape": (128, 128, 3)})
ds.process_metadata(cfg, True)
m = ConfigNetFirstStage(cfg, seed=0); m.use_graphs = True
m.setup_training(None, ds, 0, real_training_set=ds)
dopt, gopt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
def it():
    a, b, c = m.run_concurrently([lambda: m.discriminator_training_step(ds, dopt), lambda: m.synth_discriminator_training_step(ds, dopt),
                                  lambda: m.latent_discriminator_training_step(ds, dopt)])
    g = m.generator_training_step(ds, ds, gopt); m.update_smoothed_weights(); return g
for _ in range(3): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g = it()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["config0_first_stage_128_b8"] = {"images_per_sec": round(8 * 20 / dt, 2), "ms_per_iteration": round(1e3 * dt / 20, 3), "finite": bool(np.isfinite(float(g["loss_sum"])))}
print(json.dumps(out))
