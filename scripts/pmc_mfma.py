"""MFMA-busy fraction per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES run (rocpd
sqlite).  Normalisation (checked on one convolution with a known MFMA count: 18.87 M v_mfma_f32_32x32x2 x 64 cycles =
1.208e9 = the counter): SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of pipe-busy cycles; GRBM_GUI_ACTIVE is summed
over the 8 XCDs, so elapsed cycles = GUI_ACTIVE / 8 and MFMA-busy = BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).  The same
numbers give the effective clock (elapsed cycles / kernel time)."""
import glob, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernels_hash
from collections import defaultdict
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
for name, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name.replace("(anonymous namespace)::", "").replace("void ", "")
    k = k.split("(")[0][:64] if "fwd2_kernel<" in k else k[:48]      # (fwd2's ten template arguments tell its variants apart)
    acc[k][cn] += val
    if cn == "GRBM_GUI_ACTIVE": calls[k] += 1
N_SIMD, N_XCD = 256 * 4, 8
rows = []
for k, c in acc.items():
    gui, busy = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui > 0:
        rows.append((busy, k, calls[k], gui, busy / (gui / N_XCD * N_SIMD), c.get("SQ_BUSY_CU_CYCLES", 0.0)))
tot_busy = sum(r[0] for r in rows if any(t in r[1] for t in ("igemm", "fwd2_kernel", "wgrad2", "wino_fwd", "wino4_fwd", "s2_image", "s1_image", "c3_fwd", "c7s2_fwd", "c3_wgrad_kernel", "up2k4")))
tot_gui = sum(r[3] for r in rows if any(t in r[1] for t in ("igemm", "fwd2_kernel", "wgrad2", "wino_fwd", "wino4_fwd", "s2_image", "s1_image", "c3_fwd", "c7s2_fwd", "c3_wgrad_kernel", "up2k4")))
for r in sorted(rows, reverse=True)[:14]:
    print("%-66s calls %6d  GUI_ACTIVE %14.0f  MFMA_BUSY %16.0f  MFMA-busy %.3f  SQ_BUSY_CU %14.0f" % (r[1], r[2], r[3], r[0], r[4], r[5]))
out = {"kernels_hash": kernels_hash(), "kernel_class": "igemm_fwd/fwd2/igemm_wgrad/wgrad2/wino_fwd/wino4_fwd/s2_image_dgrad/s1_image_dgrad/c3_fwd/c3_wgrad/up2k4_rgb_fwd", "mfma_busy_cycles": tot_busy, "gui_active_cycles": tot_gui,
       "mfma_busy_fraction": tot_busy / (tot_gui / N_XCD * N_SIMD) if tot_gui else None,
       "normalisation": "SQ_VALU_MFMA_BUSY_CYCLES (sum over SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)"}
print(json.dumps(out))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
