#!/usr/bin/env python3
"""Build profiles/r01_summary.md and profiles/r01_pmc_traffic.json from rocprofv3 outputs.

usage: make_summary.py <kernel_stats.csv> <pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir> <batch>
  kernel_stats.csv : rocprofv3 --kernel-trace --stats ... (the *_kernel_stats.csv file)
  pmc dirs         : rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of the SAME command (separate passes)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  MI355X_MICROARCH.md: FETCH_SIZE under-counts wide (16 B/lane)
streaming reads by 2x; other access widths are uncalibrated -- the raw value is kept here and k_resize (a pure streaming
kernel with a known byte count) serves as the calibration row."""
import csv, glob, json, os, re, sys
from collections import defaultdict

stats_csv, fetch_dir, write_dir, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kname(full):
    m = re.search(r"\b(k_\w+)", full)
    return m.group(1) if m else full.split("(")[0][:40]


def pmc(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[kname(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}


rows = list(csv.DictReader(open(stats_csv)))
fetch, write = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE + "/..")
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(HERE, "..", "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
alg = bench.algorithmic_bytes_per_pair()
out = ["# Round 1 profile summary (1x MI355X, `bench.py --serial --batch %d`)" % batch,
       "Source: `rocprofv3 --kernel-trace --stats` (kernel table) and separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of",
       "the same command (rocprofv3 reports both in KB).  Per MI355X_MICROARCH.md FETCH_SIZE is calibrated (x2) only for 16-B/lane",
       "streams; these kernels mostly use 4-byte loads, so the RAW counter is listed and `k_resize` (pure streaming, known byte",
       "count) is the calibration row.  Made by profiles/probes/make_summary.py.", "",
       "## Kernel time (single-stream steps)", "", "| kernel | calls | avg us | % |", "|---|---|---|---|"]
for r in rows[:16]:
    out.append("| %s | %s | %.1f | %s |" % (kname(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
out += ["", "## HBM traffic per launch (MB, mean over the front and bird launches), batch %d" % batch, "",
        "| kernel | FETCH_SIZE | WRITE_SIZE | algorithmic bytes (read+write, mean per launch) |", "|---|---|---|---|"]
traffic = {}
launches_per_step = {"k_resize": 14, "k_proj_frame": 1, "k_bird_mappoints": 1, "k_pose_opt": 1}  # others: front + bird = 2
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    a = alg.get("k_resize" if k == "k_resize_rows" else k)
    lk = "k_resize" if k == "k_resize_rows" else k
    a_launch = a * batch / launches_per_step.get(lk, 2) if a else None
    traffic[k] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "algorithmic_bytes_per_launch": a_launch}
    out.append("| %s | %.1f | %.1f | %s |" % (k, f / 1e6, w / 1e6, "%.1f" % (a_launch / 1e6) if a_launch else "-"))
open(os.path.join(HERE, "r01_summary.md"), "w").write("\n".join(out) + "\n")
json.dump({"batch": batch, "unit": "bytes per launch (mean over launches), raw rocprofv3 FETCH_SIZE/WRITE_SIZE x 1024",
           "kernels": traffic}, open(os.path.join(HERE, "r01_pmc_traffic.json"), "w"), indent=1)
print("\n".join(out[-12:]))
