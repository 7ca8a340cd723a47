import sys, numpy as np, torch
sys.path.insert(0, ".")
from confignet_amd import ConfigNet, SyntheticFaceDataset, optim, ops
from confignet_amd import functional as F
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG, frozen
from confignet_amd.confignet_utils import merge_configs
from confignet_amd.nn import WEIGHTS_EPOCH
ds = SyntheticFaceDataset(16, 128, seed=3)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 4, "output_shape": (128, 128, 3)})
ds.process_metadata(cfg, True)
np.random.seed(5)
m = ConfigNet(cfg, seed=0)
m.setup_training(None, ds, 0, real_training_set=ds)
m._stage_synth("g", ds, 2); m._stage_real("g", ds, 2)
z = torch.randn(2, 145, device="cuda"); rot = torch.zeros(2, 3, device="cuda"); rot[:, 0] = 0.3
x5 = torch.randn(2, 4, 4, 4, 512, device="cuda"); sb = torch.randn(2, 512, device="cuda")
w3 = m.generator.weights[2]; b3 = m.generator.weights[3]
from confignet_amd.dnn_models.hologan_generator import C3_UP

def run(name, fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ref = [v.clone() for v in fn()]
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); WEIGHTS_EPOCH[0] += 1
    with torch.cuda.graph(g, stream=s):
        outs = fn()
    res = []
    for r in range(3):
        g.replay(); torch.cuda.synchronize()
        res.append(max(float((a.float() - b.float()).abs().max()) for a, b in zip(outs, ref)))
    print("%-28s replay errs %s" % (name, ["%.2e" % e for e in res]))

with torch.no_grad():
    run("gemm learned input", lambda: [F.linear(torch.zeros((2, 1), device="cuda"), m.generator.weights[0], m.generator.weights[1])])
    run("conv3d up split-K + lrelu", lambda: [F.conv(x5, w3, b3, C3_UP, 1, 0.3)])
    run("adain", lambda: [F.adain(x5, sb)])
    run("rotate", lambda: [F.rotate3d(x5.reshape(2, 8, 8, 8, 64).contiguous(), rot)])
    run("generator fwd", lambda: [m.generator((z, rot))])
    run("encoder fwd", lambda: list(m.encoder(m._real_imgs("g", ds))))
run("generator fwd+bwd", lambda: list(torch.autograd.grad(m.generator((z, rot)).sum(), m.generator.trainable_weights[1:4])))
