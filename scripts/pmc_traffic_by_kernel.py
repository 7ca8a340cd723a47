"""profiles/roundN_pmc_traffic_by_kernel.txt: measured HBM-side bytes per launch of every kernel of the convolution class
(rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `bench.py --serial --steps K`) next to the ALGORITHMIC
bytes per launch of the same kernels (every operand read once, the result written once: cn_prof_collect_by_family, printed by
the bench line of the same command as roofline.by_kernel).
usage: pmc_traffic_by_kernel.py <fetch_dir> <write_dir> <bench_line.json> <out.txt> [<out.json>]"""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernels_hash

TILES = {"2, 2, 2, 2": "128x128", "2, 2, 2, 1": "128x64", "2, 2, 1, 1": "64x64", "4, 1, 1, 1": "128x32", "4, 1, 1, 3": "128x96", "4, 1, 2, 2": "256x64"}


def family(name):
    for k in ("igemm_fwd_kernel", "igemm_wgrad_kernel"):
        if k in name:
            t = name.split(k + "<")[1][:10]
            return "%s<%s>" % (k[:-7], TILES.get(t, t))
    if "wgrad2_kernel" in name:                       # the LDS-DMA filter gradient: same tile families as igemm_wgrad_kernel
        t = name.split("wgrad2_kernel<")[1][:10]      # (its ordered reduction, sum_parts_kernel, is not a kernel of the class)
        return "igemm_wgrad<%s>" % TILES.get(t, t)
    if "wino4_fwd_kernel" in name:
        return "wino_fwd"                             # (F(2x2) and F(4x4) are one family in cn_prof_collect_by_family)
    if "fwd2_kernel<" in name:                        # the LDS-DMA main loop (round 5): fp32 -> igemm_fwd's tile families, bf16 -> igemm_bf16
        args = [a.strip() for a in name.split("fwd2_kernel<")[1].split(">")[0].split(",")]
        if args[-1] in ("true", "1"):
            return "igemm_bf16"
        return "igemm_fwd<%s>" % TILES.get(", ".join(args[:4]), ", ".join(args[:4]))
    if "s1_image_dgrad_kernel" in name:
        return "s2_image_dgrad"                       # (one family in cn_prof_collect_by_family)
    for k, f in (("wino_fwd_kernel", "wino_fwd"), ("c3_fwd_kernel", "c3_fwd"), ("c7s2_fwd_kernel", "c3_fwd"), ("s2_image_dgrad_kernel", "s2_image_dgrad"),
                 ("c3_wgrad_kernel", "c3_wgrad"), ("up2k4_rgb_fwd_kernel", "thin / up2k4_rgb"), ("igemm_bf16_wgrad_tr_kernel", "igemm_bf16_wgrad"), ("igemm_bf16_wgrad_kernel", "igemm_bf16_wgrad"),
                 ("igemm_bf16_kernel", "igemm_bf16")):
        if k in name:
            return f
    return None


def load(d, counter):
    cur = sqlite3.connect(glob.glob(d + "/**/*.db", recursive=True)[0]).cursor()
    acc = defaultdict(lambda: [0, 0.0])
    for name, val in cur.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
        f = family(name)
        if f:
            acc[f][0] += 1
            acc[f][1] += val * 1024.0            # KiB
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
line = json.loads([l for l in open(sys.argv[3]).read().splitlines() if l.startswith("{")][-1])
model = line["roofline"]["by_kernel"]
rows, tot_m, tot_a = [], 0.0, 0.0
for f in sorted(set(fetch) | set(write), key=lambda k: -(fetch[k][1] * 2 + write[k][1])):
    nf, fb = fetch[f]
    nw, wb = write[f]
    meas = 2 * fb / max(nf, 1) + wb / max(nw, 1)
    alg = model.get(f, {}).get("algorithmic_mb_per_launch")
    rows.append((f, nf, 2 * fb / max(nf, 1) / 1e6, wb / max(nw, 1) / 1e6, meas / 1e6, alg, (meas / 1e6 / alg) if alg else None))
    if alg:
        tot_m += meas * nf
        tot_a += alg * 1e6 * nf
with open(sys.argv[4], "w") as fp:
    fp.write("# kernels_hash %s; command: %s\n" % (kernels_hash(), line.get("command", "python bench.py --serial")))
    fp.write("# HBM-side bytes per launch: FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half of wide coalesced reads;\n"
             "# scalar / narrow accesses are not under-reported, so the doubled figure is an UPPER bound for kernels with such\n"
             "# loads) + WRITE_SIZE (uncalibrated); algorithmic = x, w read once + y written once (cn_prof_collect_by_family).\n"
             "# Infinity-Cache hits are counted by these counters; tensors under ~100 MB that a previous kernel just wrote do not\n"
             "# come from HBM, so 'measured' is fabric traffic, an upper bound of HBM traffic.\n")
    fp.write("%-24s %8s %12s %12s %12s %14s %8s\n" % ("kernel", "launches", "fetch x2 MB", "write MB", "measured MB", "algorithmic MB", "ratio"))
    for f, n, fm, wm, mm, alg, ratio in rows:
        fp.write("%-24s %8d %12.2f %12.2f %12.2f %14s %8s\n" % (f, n, fm, wm, mm, "%.2f" % alg if alg else "-", "%.2f" % ratio if ratio else "-"))
    if tot_a:
        fp.write("%-24s %8s %12s %12s %12s %14s %8.2f   (launch-weighted, kernels with a byte model)\n" % ("class", "", "", "", "", "", tot_m / tot_a))
print(open(sys.argv[4]).read())
if len(sys.argv) > 5:
    json.dump({"kernels_hash": kernels_hash(), "measured_over_algorithmic": tot_m / tot_a if tot_a else None,
               "by_kernel": {f: {"launches": n, "fetch_x2_mb": fm, "write_mb": wm, "algorithmic_mb": alg, "ratio": ratio}
                             for f, n, fm, wm, mm, alg, ratio in rows}}, open(sys.argv[5], "w"), indent=1)
