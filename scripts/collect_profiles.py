#!/usr/bin/env python3
"""Copy the evidence set that scripts/run_profile.sh and scripts/run_pmc_ivf.sh left under gpurun_out/ into profiles/
under a tag (e.g. r03_i), and refresh profiles/traffic.json from the PMC summaries.
usage: collect_profiles.py TAG"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = os.path.join(root, "gpurun_out")
p = os.path.join(root, "profiles")


def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def put(src, name):
    shutil.copyfile(src, os.path.join(p, f"{tag}_{name}"))


for src, name in (("prof/bench.json", "bench.json"), ("prof/bench_traced.json", "bench_traced.json")):
    with open(os.path.join(p, f"{tag}_{name}"), "w") as f:
        json.dump(last_json_line(os.path.join(g, src)), f, indent=1)
put(os.path.join(g, "prof/kernel_stats_rag.csv"), "bench_kernel_stats.csv")
trace = glob.glob(os.path.join(g, "prof/trace/*/*kernel_trace.csv"))[0]
with open(os.path.join(p, f"{tag}_bench_kernel_launch_groups.csv"), "w") as f:
    subprocess.run([sys.executable, os.path.join(root, "scripts/launch_groups.py"), trace], stdout=f, check=True)
put(os.path.join(g, "pmc/summary_b1.csv"), "pmc_batch1.csv")
put(os.path.join(g, "pmc/summary_b4096.csv"), "pmc_batch4096.csv")
if os.path.exists(os.path.join(g, "pmc_ivf/summary.csv")):
    put(os.path.join(g, "pmc_ivf/summary.csv"), "ivf_pmc.csv")
    put(os.path.join(g, "pmc_ivf/kernel_stats_rag.csv"), "ivf_kernel_stats.csv")
    with open(os.path.join(p, f"{tag}_ivf_traced.json"), "w") as f:
        json.dump(last_json_line(os.path.join(g, "pmc_ivf/traced.json")), f, indent=1)


def counters(path):
    out = {}
    with open(path) as f:
        for row in csv.reader(f):
            if len(row) >= 5:
                out[(row[1], row[2])] = row
    return out


b1 = counters(os.path.join(g, "pmc/summary_b1.csv"))
b4 = counters(os.path.join(g, "pmc/summary_b4096.csv"))


def pick(c, kernel_prefix, counter, col):
    for (kern, cn), row in c.items():
        if kern.replace("void ", "").startswith(kernel_prefix) and cn == counter:
            return float(row[col])
    raise KeyError((kernel_prefix, counter))


scan = "rag::flat_scan_kernel<64, 3, 4, 1"
gemm = "rag::gemm_topk_kernel<256, 256, 8, false, false, false>"
scan_bytes = pick(b1, scan, "FETCH_SIZE", 5) + pick(b1, scan, "WRITE_SIZE", 5)
gemm_bytes = pick(b4, gemm, "FETCH_SIZE", 5) + pick(b4, gemm, "WRITE_SIZE", 5)
mfma = pick(b4, gemm, "SQ_VALU_MFMA_BUSY_CYCLES", 4)
gui = pick(b4, gemm, "GRBM_GUI_ACTIVE", 4)
traffic = {
    "rows": 10000000, "dim": 768,
    "flat_scan_kernel": {"batch": 1, "hbm_bytes_per_launch": int(scan_bytes),
                         "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/run_pmc.sh), FETCH_SIZE in KiB "
                                   f"doubled per the gfx950 correction; profiles/{tag}_pmc_batch1.csv"},
    "gemm_topk_kernel": {"batch": 4096, "hbm_bytes_per_launch": int(gemm_bytes),
                         "source": f"same passes at batch 4096 (256x256 tile shape); profiles/{tag}_pmc_batch4096.csv. FETCH_SIZE counts "
                                   "L2-to-fabric requests, Infinity Cache hits included",
                         "mfma_util": {"SQ_VALU_MFMA_BUSY_CYCLES": mfma, "GRBM_GUI_ACTIVE_sum_over_8_XCDs": gui, "simds": 1024,
                                       "util": mfma / (gui / 8 * 1024)}},
}
with open(os.path.join(p, "traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
print(json.dumps(traffic, indent=1))
