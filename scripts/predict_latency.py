"""Latency of model.generate_images (the demo / predict path) at a few batch sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from confignet_amd import ConfigNet, SyntheticFaceDataset
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG
from confignet_amd.confignet_utils import merge_configs
ds = SyntheticFaceDataset(8, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 16, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0)
for graphs in (False, True):
    m.use_inference_graphs = graphs
    for n in (1, 6, 32):
        lat = m.sample_latent_vector(n); rot = m.sample_rotations(n)
        for _ in range(3): m.generate_images(lat, rot)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): m.generate_images(lat, rot)
        torch.cuda.synchronize()
        print("generate_images N=%d (%s): %.3f ms per call incl. the uint8 copy to the host" % (n, "replayed HIP graph" if graphs else "eager dispatch", (time.perf_counter() - t) / 50 * 1e3))
