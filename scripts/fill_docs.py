"""Fill the @@PLACEHOLDER@@ fields of DESIGN.md / README.md from the committed profile set (profiles/round6_*):
    python scripts/fill_docs.py            # prints the values; --write replaces them in the documents"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def line(name):
    return json.loads(open(os.path.join(P, name)).read().strip().splitlines()[-1])


b = line("round6_bench.json")
r = b["roofline"]
sec = b["secondary"]
cfgs = sec["configs"]
trace = open(os.path.join(P, "round6_bench_serial_kernel_trace.txt")).read()
tot = int(re.search(r"over (\d+) dispatches", trace).group(1))
adam = sum(int(m.group(1)) for m in re.finditer(r"adam_kernel.*?\s(\d+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s*$", trace, re.M))
traffic = json.load(open(os.path.join(P, "round6_pmc_traffic.json")))
mfma = json.load(open(os.path.join(P, "round6_pmc_mfma.json")))
alg_per_launch = r["algorithmic_bytes_per_step"] / r["launches_per_step"]
vals = {
    "FP32": "%.1f" % b["value"], "FP32MS": "%.2f" % b["ms_per_step"],
    "FP32_400": "%.1f" % line("round6_bench_400_steps.json")["value"], "FP32_2000": "%.1f" % line("round6_bench_2000_steps.json")["value"],
    "PARITY": "%.1e" % b["loss_parity_vs_cpu"]["max_err"],
    "BF16": "%.1f" % sec["bf16"]["value"], "BF16MS": "%.2f" % sec["bf16"]["ms_per_step"],
    "BF16_400": "%.1f" % line("round6_bench_bf16_400_steps.json")["value"],
    "CLASSMS": "%.1f" % r["kernel_ms_per_step"], "TFLOPS": "%.1f" % r["achieved"], "FRAC": "%.3f" % r["frac"],
    "BUSY": "%.3f" % mfma["mfma_busy_fraction"], "TRAFFIC": "%.1f" % (traffic["hbm_bytes_per_launch"] / 1e6),
    "TRATIO": "%.2f" % (traffic["hbm_bytes_per_launch"] / alg_per_launch),
    "FT": "%.2f" % cfgs["config4_finetune_256_1img_200steps"]["seconds"], "FTL": "%.2f" % cfgs["config4_finetune_256_1img_200steps_literal"]["seconds"],
    "LG": "%.1f" % cfgs["config5_latentgan_b4096"]["ms_per_step"], "LGD": "%.2f" % cfgs["config5_latentgan_b4096_device_sampling"]["ms_per_step"],
    "FS": "%.0f" % cfgs["config0_first_stage_128_b8"]["images_per_sec"], "CPU": "%.2f" % b["cpu_baseline"]["value"],
    "LAUNCHES": "{:,}".format(int(round(tot / (adam / 7.0)))).replace(",", " "),
    "GFLOP": "{:,}".format(int(round(r["algorithmic_gflop_per_step"]))).replace(",", " "),
    "MFMAMS": "%.1f" % (r["algorithmic_gflop_per_step"] / 157.3), "PIPEFRAC": "%.2f" % r["pipelined"]["frac"],
    "SPEEDUP": "%.1f" % (100.0 * (b["value"] / 396.2 - 1.0)),
    "FP32_VERDICT": ("at the target on the profile set's box; 421 - 426 over the boxes of the round's last hours" if b["value"] >= 425.0 else
                     "%.1f short on the profile set's box; 421 - 426 over the boxes of the round's last hours" % (425.0 - b["value"])),
}
for k, v in vals.items():
    print("%-14s %s" % (k, v))
if "--write" in sys.argv:
    for doc in ("DESIGN.md", "README.md"):
        path = os.path.join(ROOT, doc)
        text = open(path).read()
        for k, v in vals.items():
            text = text.replace("@@%s@@" % k, v)
        left = sorted(set(re.findall(r"@@[A-Z0-9_]+@@", text)))
        if left:
            print("%s: unfilled %s" % (doc, left))
        open(path, "w").write(text)
