"""Dev: which call sites the elementwise / reduction passes of one eager iteration come from, and how many bytes they touch
(tensor arguments' sizes summed per call): python scripts/dev/ew_sites.py"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from confignet_amd import ops
model, real_set, synth_set, d_opt, g_opt, cfg = bench.setup(16, 256, 64)
model.use_graphs = False
for _ in range(2):
    model.training_iteration(real_set, synth_set, d_opt, g_opt)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0])
NAMES = ["nc_reduce", "nc_reduce4", "nc_lin2", "act_bwd", "act_bwd_partials", "act_fwd", "nc_reduce_dact", "dual_tail_gx", "dual_tail_gx_tx", "nc_reduce_hxt",
         "norm_apply_fwd", "norm_apply_bwd", "maxpool2_bwd_relu", "bn_act_bwd", "maxpool_fwd", "maxpool_bwd", "sqdiff_sum", "row_scale_diff", "tap_bwd", "row_scale",
         "axpby", "mul", "sumpool2", "masked_diff", "zero_"]
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        out = fn(*a, **k)
        nb = 0
        for t in list(a) + list(k.values()) + (list(out) if isinstance(out, (tuple, list)) else [out]):
            if torch.is_tensor(t):
                nb += t.numel() * t.element_size()
        st = traceback.extract_stack(limit=6)
        site = " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st[:-1]) if "ew_sites" not in f.filename and "autograd" not in f.filename)[:90]
        e = agg[(name, site)]
        e[0] += 1
        e[1] += nb
        return out
    setattr(ops, name, w)
for n in NAMES:
    wrap(n)
model.training_iteration(real_set, synth_set, d_opt, g_opt)
torch.cuda.synchronize()
tot = sum(v[1] for v in agg.values())
print("total %.2f GB in %d calls" % (tot / 1e9, sum(v[0] for v in agg.values())))
for (name, site), (c, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[1]) if len(sys.argv) > 1 else 45]:
    print("%7.1f MB %4d  %-18s %s" % (nb / 1e6, c, name, site))
