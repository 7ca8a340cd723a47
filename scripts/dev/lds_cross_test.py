"""Does a kernel with LDS-DMA disturb a co-resident workgroup of another kernel?  Victim launches on one stream are compared
bit for bit with their serial result while aggressor launches run on a second stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ops

torch.manual_seed(0)
dev = "cuda"


def rnd(*s):
    return torch.randn(*s, device=dev) * 0.05


def victim_dgrad_s2(n=8, hw=64, cin=96, cout=192):          # parity-ordered data gradient (igemm_fwd, par)
    g = ops.ConvSpec((3, 3), stride=2).geom((n, hw, hw, cin), cout)
    gy, w = rnd(*ops.geom_out_shape(g)), rnd(3, 3, cin, cout)
    return lambda: ops.conv_dgrad(gy, w, g)


def victim_fwd(n=8, hw=64, cin=64, cout=256, k=3):
    g = ops.ConvSpec((k, k)).geom((n, hw, hw, cin), cout)
    x, w, b = rnd(n, hw, hw, cin), rnd(k, k, cin, cout), rnd(cout)
    os.environ["CN_NO_WINOGRAD"] = "1"
    return lambda: ops.conv_fwd(x, w, b, g)


def aggr_wgrad(n, hw, cin, cout, stride=1, k=3):
    g = ops.ConvSpec((k, k), stride=stride).geom((n, hw, hw, cin), cout)
    x, gy = rnd(n, hw, hw, cin), rnd(*ops.geom_out_shape(g))
    return lambda: ops.conv_wgrad(x, gy, g, (k, k, cin, cout))


def aggr_fwd(n, hw, cin, cout):
    g = ops.ConvSpec((3, 3)).geom((n, hw, hw, cin), cout)
    x, w, b = rnd(n, hw, hw, cin), rnd(3, 3, cin, cout), rnd(cout)
    return lambda: ops.conv_fwd(x, w, b, g)


ops.WINOGRAD = False
victims = {"dgrad_s2_96_192": victim_dgrad_s2(), "dgrad_s2_48_96": victim_dgrad_s2(8, 128, 48, 96),
           "dgrad_s2_192_384": victim_dgrad_s2(16, 32, 192, 384), "fwd3x3_64_256": victim_fwd()}
ops.WINOGRAD = True
aggressors = {"wgrad2_128x128": aggr_wgrad(8, 32, 192, 384, 2), "wgrad2_64x64": aggr_wgrad(8, 16, 256, 256),
              "wgrad2_96": aggr_wgrad(8, 64, 96, 192, 2), "wgrad2_small": aggr_wgrad(8, 8, 512, 512),
              "wino2": aggr_fwd(8, 32, 512, 512), "wino4": aggr_fwd(8, 64, 256, 256), "none": None}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for vn, v in victims.items():
    ops.WINOGRAD = False
    ref = v().clone()
    torch.cuda.synchronize()
    for an, a in aggressors.items():
        ops.WINOGRAD = True
        bad = 0
        for rep in range(60):
            with torch.cuda.stream(sb):
                if a is not None:
                    for _ in range(6):
                        a()
            with torch.cuda.stream(sa):
                ops.WINOGRAD = False
                outs = [v() for _ in range(4)]
                ops.WINOGRAD = True
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(o, ref)) for o in outs)
        print("victim %-18s aggressor %-16s mismatching launches %d / 240" % (vn, an, bad), flush=True)
