"""Device-side timeline of the pipelined iterations: events at the start of the discriminator phase, at its join, after
the generator tail and after the EMA, over K iterations without host synchronisation."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from confignet_amd import ConfigNet, SyntheticFaceDataset, optim
from confignet_amd.confignet_first_stage import DEFAULT_CONFIG, ConfigNetFirstStage
from confignet_amd.confignet_utils import merge_configs
np.random.seed(0)
ds = SyntheticFaceDataset(512, 256, seed=1)
cfg = merge_configs(DEFAULT_CONFIG, {"batch_size": 16, "output_shape": (256, 256, 3)})
ds.process_metadata(cfg, True)
m = ConfigNet(cfg, seed=0); m.use_graphs = True; m.overlap_discriminators = os.environ.get('CN_NO_D_OVERLAP') is None
m.setup_training(None, ds, 0, real_training_set=ds)
dopt, gopt = optim.Adam(**cfg["optimizer"]), optim.Adam(**cfg["optimizer"])
for _ in range(5):
    m.training_iteration(ds, ds, dopt, gopt)
torch.cuda.synchronize()
marks = []
line_marks = []
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
orig = ConfigNetFirstStage._flush_deferred
def flush(self):
    pending, self._deferred = self._deferred, []
    if not pending:
        return
    follower = pending.pop() if (self._deferred_then and len(pending) > 1 and pending[-1].early_cut) else None
    cur = torch.cuda.current_stream()
    a = ev()
    lines = {}
    for g in pending:
        g.stream.wait_stream(cur)
        with torch.cuda.stream(g.stream):
            if getattr(g, "prelaunched", False):
                g.replay(g.early_cut); g.prelaunched = False
            else:
                g.replay()
            g.finish()
            lines[getattr(g, "name", "?")] = ev()
    ev_early = None
    if follower is not None and self.early_generator_forward:
        follower.replay(0, follower.early_cut)
        lines["g_early"] = ev()
        ev_early = torch.cuda.Event()
        ev_early.record(cur)
    for g in pending:
        cur.wait_stream(g.stream)
    b = ev()
    if follower is not None:
        follower.replay(follower.early_cut if self.early_generator_forward else 0)
        follower.finish()
        if self.overlap_discriminators:
            for g in pending:
                st = self._stagers.get(getattr(g, "name", None))
                if st is None or not g.early_cut:
                    continue
                stage, training_set, optimizer = st
                with torch.cuda.stream(g.stream):
                    stage(training_set)
                    g.replay(0, g.early_cut)
                    lines["real_" + g.name] = ev()
                g.prelaunched = True
                net = self.discriminator if g.name == "d" else self.synth_discriminator
                self._prestaged[g.name] = (id(training_set), id(optimizer), net.epoch, self._bufs.generation)
            self._prelaunch_generator_targets(pending, follower, ev_early)
            if self._targets_ahead is not None:
                with torch.cuda.stream(self._targets_ahead[2]):
                    lines["targets"] = ev()
    c = ev()
    marks.append([a, b, c])
    line_marks.append(lines)
ConfigNetFirstStage._flush_deferred = flush
K = 12
for _ in range(K):
    m.training_iteration(ds, ds, dopt, gopt)
    marks[-1].append(ev())
torch.cuda.synchronize()
t0 = marks[0][0]
for i, (a, b, c, d) in enumerate(marks):
    nxt = marks[i + 1][0] if i + 1 < len(marks) else None
    print("iter %2d: start %8.2f  d_phase %6.2f  g_tail %6.2f  ema+rest %5.2f  gap to next start %5.2f" % (
        i, t0.elapsed_time(a), a.elapsed_time(b), b.elapsed_time(c), c.elapsed_time(d), d.elapsed_time(nxt) if nxt else 0.0))
    print("         lines (ms after the iteration's start): " + "  ".join("%s %.2f" % (k, a.elapsed_time(e)) for k, e in line_marks[i].items()))
