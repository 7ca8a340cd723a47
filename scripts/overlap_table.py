"""profiles/roundN_overlap_table.txt: one row per kernel of the convolution class -- launches, average duration, MFMA-busy fraction,
the time its matrix pipes were busy (t_mfma) and the rest (t_rest = duration - t_mfma: operand delivery, ramp / tail, barriers,
epilogue that did NOT overlap the MFMAs), and the measured / algorithmic traffic ratio of its family where the byte model has one.
Inputs: the rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (+ --kernel-trace) over `bench.py --serial`, and the
by-kernel traffic json of scripts/pmc_traffic_by_kernel.py.
usage: overlap_table.py <mfma_pass_dir> <traffic_by_kernel.json> <out.txt>"""
import glob, json, os, sqlite3, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
spec = importlib.util.spec_from_file_location("tbk", os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic_by_kernel.py"))
from bench import kernels_hash

TILES = {"2, 2, 2, 2": "128x128", "2, 2, 2, 1": "128x64", "2, 2, 1, 1": "64x64", "4, 1, 1, 1": "128x32", "4, 1, 1, 3": "128x96"}


def family(name):          # (the family naming of pmc_traffic_by_kernel.py)
    if "fwd2_kernel<" in name:
        args = [a.strip() for a in name.split("fwd2_kernel<")[1].split(">")[0].split(",")]
        return "igemm_bf16" if args[-1] in ("true", "1") else "igemm_fwd<%s>" % TILES.get(", ".join(args[:4]), "?")
    for k in ("igemm_fwd_kernel", "igemm_wgrad_kernel", "wgrad2_kernel"):
        if k + "<" in name:
            t = name.split(k + "<")[1][:10]
            return ("igemm_wgrad<%s>" if "wgrad" in k else "igemm_fwd<%s>") % TILES.get(t, t)
    for k, f in (("wino4_fwd_kernel", "wino_fwd"), ("wino_fwd_kernel", "wino_fwd"), ("c3_fwd_kernel", "c3_fwd"), ("c7s2_fwd_kernel", "c3_fwd"), ("s2_image_dgrad_kernel", "s2_image_dgrad"),
                 ("s1_image_dgrad_kernel", "s2_image_dgrad"), ("c3_wgrad_kernel", "c3_wgrad"), ("up2k4_rgb_fwd_kernel", "thin / up2k4_rgb"),
                 ("igemm_bf16_wgrad", "igemm_bf16_wgrad"), ("igemm_bf16_kernel", "igemm_bf16")):
        if k in name:
            return f
    return None


db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
cur = con.cursor()
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
for name, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    if family(name) is None:
        continue
    k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][cn] += val
    if cn == "GRBM_GUI_ACTIVE":
        calls[k] += 1
# durations from the same run's kernel trace
dur = defaultdict(float); ndur = defaultdict(int)
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
try:
    for name, d in cur.execute("select name, (end - start) from kernels"):
        k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k in acc:
            dur[k] += d; ndur[k] += 1
except sqlite3.Error:
    pass
traffic = json.load(open(sys.argv[2]))["by_kernel"] if os.path.exists(sys.argv[2]) else {}
rows = []
for k, c in acc.items():
    gui, busy = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0:
        continue
    frac = busy / (gui / 8 * 1024)
    us = dur[k] / max(ndur[k], 1) / 1e3 if ndur[k] else gui / 8 / calls[k] / 2.4e3      # (no trace table: elapsed cycles at 2.4 GHz)
    rows.append((us * calls[k], k, calls[k], us, frac, us * frac, us * (1 - frac), (traffic.get(family(k)) or {}).get("ratio")))
with open(sys.argv[3], "w") as fp:
    fp.write("# kernels_hash %s; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -- python bench.py --serial --steps 2 --warmup 1\n" % kernels_hash())
    fp.write("# t_mfma = duration x MFMA-busy; t_rest = duration - t_mfma (what did not overlap the matrix pipe); traffic = measured / algorithmic bytes of the kernel's family\n")
    fp.write("%-62s %8s %9s %9s %9s %9s %8s %8s\n" % ("kernel", "launches", "us", "MFMA-busy", "t_mfma us", "t_rest us", "traffic", "share"))
    tot = sum(r[0] for r in rows)
    for r in sorted(rows, reverse=True):
        fp.write("%-62s %8d %9.1f %9.3f %9.1f %9.1f %8s %7.1f%%\n" % (r[1][:62], r[2], r[3], r[4], r[5], r[6], "%.2f" % r[7] if r[7] else "-", 100 * r[0] / tot))
    tm = sum(r[5] * r[2] for r in rows); tr = sum(r[6] * r[2] for r in rows)
    fp.write("%-62s %8d %9s %9.3f %9.0f %9.0f   (sums over the sampled launches, us)\n" % ("class", sum(r[2] for r in rows), "", tm / (tm + tr), tm, tr))
print(open(sys.argv[3]).read())
