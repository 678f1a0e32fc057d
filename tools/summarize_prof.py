#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the tracked summaries under profiles/.

usage: tools/summarize_prof.py <round-tag> <kernel_stats.csv> <fetch_counter.csv> <write_counter.csv> [bench.json]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of coalesced
streaming reads (MI355X_MICROARCH.md "HBM"; confirmed for this code's 8 B/lane pattern on
k_linearize_between, whose unique input bytes are known), so read bytes = 2 * FETCH_SIZE * 1024.
"""
import collections
import csv
import json
import os
import shutil
import sys

tag, stats, fetch, write = sys.argv[1:5]
bench = sys.argv[5] if len(sys.argv) > 5 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in d.items()}


f, w = agg(fetch), agg(write)
dur = {r["Name"].split("(")[0]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(stats))}
lines = [f"# {tag}: rocprofv3 PMC summary (per launch, averaged over the dispatches of the run)", "",
         "| kernel | launches (trace run) | avg duration ms | read GB = 2*FETCH_SIZE*1024 | write GB = WRITE_SIZE*1024 | HBM GB/s |",
         "|---|---|---|---|---|---|"]
for k in sorted(dur, key=lambda k: -dur[k][0] * dur[k][1]):
    if not k.startswith("vf::"):
        continue
    rd = 2 * f.get(k, (0, 0))[1] * 1024 / 1e9
    wr = w.get(k, (0, 0))[1] * 1024 / 1e9
    ms = dur[k][1] / 1e6
    lines.append(f"| {k} | {dur[k][0]} | {ms:.4f} | {rd:.4f} | {wr:.4f} | {(rd + wr) / (ms / 1e3):.0f} |")
if bench:
    b = json.load(open(bench))
    lines += ["", "bench.py line of the same configuration (un-profiled run):", "", "```json", json.dumps(b), "```"]
open(os.path.join(out, f"{tag}_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
# per-factor HBM traffic of K1 for bench.py's roofline.traffic (needs the factor count of the run)
if bench:
    n_imu = json.load(open(bench))["config"]["factors_per_gpu"]["imu"]
    k = "vf::k_linearize_imu"
    per = (2 * f[k][1] + w[k][1]) * 1024 / n_imu
    json.dump({"k1_bytes_per_imu_factor": per, "read_bytes_per_imu_factor": 2 * f[k][1] * 1024 / n_imu,
               "write_bytes_per_imu_factor": w[k][1] * 1024 / n_imu,
               "source": f"profiles/{tag}_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                         "read = 2*FETCH_SIZE*1024 per MI355X_MICROARCH.md, write = WRITE_SIZE*1024)"},
              open(os.path.join(out, "traffic.json"), "w"), indent=1)
print("\n".join(lines[:16]))
