"""Training-time metrics on the HIP path: InceptionV3 features against the float64 oracle, the
KID / FID object, and the periodic checkpoint of a training run."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_metrics as M
from oracle import ref_ops as O


def t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def test_avgpool3_same_counts_only_cells_inside_the_image():
    from confignet_amd import ops
    rng = np.random.default_rng(0)
    for shape in [(2, 7, 9, 8), (1, 5, 5, 3), (3, 17, 17, 768)]:
        x = rng.normal(size=shape).astype(np.float32)
        got = ops.avgpool3_same(torch.tensor(x, device="cuda")).cpu().double()
        ref = M._avgpool3_same(t64(x))
        assert float((got - ref).abs().max()) < 1e-5


def _randomized_inception(seed=0):
    from confignet_amd.metrics.inception_distance import InceptionV3
    net = InceptionV3(rng=np.random.default_rng(seed))
    rng = np.random.default_rng(seed + 1)
    ws = net.get_weights()
    for i, (w, entry) in enumerate(zip(ws, net._entries)):   # non-trivial inference statistics: beta, mean ~ N(0, 0.1), variance in [0.5, 1.5]
        if w.ndim == 1:
            ws[i] = (rng.uniform(0.5, 1.5, size=w.shape) if entry[0].endswith("_var") else rng.normal(size=w.shape) * 0.1).astype(np.float32)
    net.set_weights(ws)
    return net, ws


def test_inception_v3_features_against_the_oracle():
    net, ws = _randomized_inception()
    assert [tuple(w.shape) for w in ws] == M.inception_weight_shapes()            # Keras get_weights() order and shapes
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, size=(2, 139, 107, 3)).astype(np.float32)             # odd extents: every VALID / SAME / pool edge rule
    got = net(x).cpu().double()
    ref = M.inception_features([t64(w) for w in ws], t64(x))
    assert got.shape == (2, 2048)
    rel = float((got - ref).norm() / ref.norm())
    assert rel < 1e-4 and float((got - ref).abs().max()) < 1e-3 * float(ref.abs().max()), rel


def test_feature_extractor_chunks_and_keras_weight_file(tmp_path):
    from confignet_amd.metrics import InceptionFeatureExtractor
    from confignet_amd.metrics.inception_distance import preprocess_input
    net, ws = _randomized_inception(5)
    path = os.path.join(tmp_path, "inception_v3_notop.npz")
    np.savez(path, *ws)                                                  # what np.savez(path, *model.get_weights()) writes
    ext = InceptionFeatureExtractor((96, 96, 3), weights_path=path)
    assert ext.model.pretrained
    imgs = np.random.default_rng(1).integers(0, 256, size=(37, 96, 96, 3)).astype(np.uint8)
    feats = ext.get_features(imgs, max_chunk_size=16, batch_size=5)     # ragged chunks and batches
    direct = net(preprocess_input(imgs[20:23])).cpu().numpy()
    np.testing.assert_allclose(feats[20:23], direct, rtol=1e-4, atol=1e-5)
    assert np.isfinite(feats).all() and feats.shape == (37, 2048)


def test_training_run_writes_metrics_images_and_checkpoints(tmp_path):
    from confignet_amd import ConfigNetFirstStage, SyntheticFaceDataset
    from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
    from confignet_amd.confignet_utils import merge_configs
    np.random.seed(0)
    ds = SyntheticFaceDataset(48, 128, seed=1)
    cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "output_shape": (128, 128, 3), "metrics_checkpoint_period": 2,
                                         "image_checkpoint_period": 2})
    ds.process_metadata(cfg, True)
    m = ConfigNetFirstStage(cfg, seed=0)
    out = str(tmp_path)
    m.train(ds, ds, out, os.path.join(out, "log"), n_steps=3, n_samples_for_metrics=24)
    # (the reference checkpoints at step 0 too: confignet_first_stage.py:349, step_number % period == 0)
    assert m.metrics["training_step_number"] == [0, 2] and len(m.metrics["kid"]) == 2 and len(m.metrics["fid"]) == 2
    assert np.isfinite(m.metrics["kid"]).all() and np.isfinite(m.metrics["fid"]).all()
    rows = np.loadtxt(os.path.join(out, "inception_metrics.txt"), ndmin=2)
    assert rows.shape == (2, 3) and list(rows[:, 0]) == [0, 2]
    files = os.listdir(os.path.join(out, "output_imgs"))
    assert any(f.startswith("000002") and "_synth" in f for f in files) and any(f.startswith("000002.png") for f in files)
    assert os.path.exists(os.path.join(out, "checkpoints", "000002.json"))
    assert os.path.exists(os.path.join(out, "generator_losses.txt"))
    # the metric object: identical image sets give FID ~ 0 through the whole device + host pipeline
    im = m._inception_metric_object
    imgs = np.asarray(ds.imgs.cpu() if torch.is_tensor(ds.imgs) else ds.imgs)[:24]
    f = im.inception_feature_extractor.get_features(imgs)
    from confignet_amd.metrics import compute_FID
    assert abs(compute_FID(f, f)) < 1e-3 * max(1.0, float(np.trace(np.cov(f, rowvar=False))))
