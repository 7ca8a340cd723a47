"""Diagnostic: fine_tune_on_img, product vs oracle after 1 and 2 iterations: per-tensor agreement of the Adam steps."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as MG
from oracle import ref_steps as S
import test_steps_gpu as T

perturb = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
m, W, inp = T._fine_tune_model()
if perturb:
    rng = np.random.default_rng(99)
    W["generator_smoothed"][1] = (1.0 + perturb * rng.standard_normal(32768)).astype(np.float32)
    m.generator_smoothed.set_weights(W["generator_smoothed"])
imgs = inp["ft_imgs"].astype(np.float32)
Wt = {k: [T.t64(w) for w in v] for k, v in W.items()}
vgg = [T.t64(w) for w in m.perceptual_loss._pretrained_dnn_activations.get_weights()]
vf = [T.t64(w) for w in m.perceptual_loss_face_reco._pretrained_dnn_activations.get_weights()]
for n_it in (1, 2):
    emb_r, rot_r, hist, gen_r = S.fine_tune_on_img(Wt, MG.FT_CFG, T.t64(imgs), n_it, vgg, vf, MG.FT_EXPR)
    m.fine_tune_loss_log = []
    emb, rot = m.fine_tune_on_img(imgs, n_iters=n_it)
    print("iters", n_it, "rot", rot, rot_r.numpy(), "emb err", np.abs(emb - emb_r.numpy()).max())
    print(" loss err", {k: round(m.fine_tune_loss_log[-1][k] - hist[-1][k], 5) for k in hist[-1]})
    for i, (a, r, w0) in enumerate(zip(m.generator_fine_tuned.get_weights(), gen_r, W["generator_smoothed"])):
        sa, sr = torch.as_tensor(a).double() - T.t64(w0), r - T.t64(w0)
        bad = ((sa - sr).abs() > 0.2e-4).double().mean()
        zero_ref = (sr.abs() < 0.5e-4).double().mean()
        print("  w[%2d] %-22s mismatch %.4f  ref-unmoved %.4f" % (i, tuple(a.shape), float(bad), float(zero_ref)))
