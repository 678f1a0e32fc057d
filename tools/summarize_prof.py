#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the tracked summaries under profiles/.

usage: tools/summarize_prof.py <round-tag> <prof-dir>      (prof-dir = gpurun_out/prof_<tag>, written by tools/profile_round.sh)

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --stats, verbatim), <tag>_pmc_summary.md, <tag>_step_timeline.txt,
profiles/kernel_durations.json (what bench.py carries in its line as `profiled_kernels`) and profiles/traffic.json.

* Durations: per kernel the plain average over all dispatches of the trace AND the average over its FULL launches --
  dispatches lasting at least half as long as the kernel's longest one.  K3 (k_assemble) is skipped for windows whose last
  LM trial was rejected, so its plain average mixes full, partial and idle launches; the full-launch figure is the one
  comparable with a stage timing.
* Traffic: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of coalesced streaming reads
  (MI355X_MICROARCH.md "HBM"; confirmed for this code's 8 B/lane pattern on k_linearize_between, whose unique input bytes
  are known), so read bytes = 2 * FETCH_SIZE * 1024.  Per kernel: mean over the larger half of its dispatches (= full
  launches), from separate --pmc passes.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag, prof = sys.argv[1:3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(prof, pattern), recursive=True)
    return f[0] if f else None


stats, trace = one("trace/**/*kernel_stats.csv"), one("trace/**/*kernel_trace.csv")
shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
short = lambda n: n.split("(")[0]

dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
kern = {}
for k, v in dur.items():
    if not k.startswith("vf::"):
        continue
    full = [x for x in v if x >= 0.5 * max(v)]
    kern[k] = {"calls": len(v), "avg_ms": sum(v) / len(v), "full_launches": len(full), "full_avg_ms": sum(full) / len(full),
               "min_ms": min(v), "max_ms": max(v)}


def pmc(sub):
    d = collections.defaultdict(list)
    f = one(f"{sub}/**/*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(sorted(v)[len(v) // 2:]) / len(sorted(v)[len(v) // 2:]) for k, v in d.items()}


fetch, write = pmc("fetch"), pmc("write")
bench = None
for name in ("bench_traced.json", "bench_default.json"):
    p = os.path.join(prof, name)
    if bench is None and os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
        bench = json.loads(lines[-1]) if lines else None
lines = [f"# {tag}: rocprofv3 summary of `python3 bench.py --steps 10 --warmup 2` (kernel trace) and its PMC passes", "",
         "| kernel | launches | avg ms | full launches | full-launch avg ms | read GB = 2*FETCH_SIZE*1024 | write GB = WRITE_SIZE*1024 | HBM GB/s (full launch) |",
         "|---|---|---|---|---|---|---|---|"]
for k in sorted(kern, key=lambda k: -kern[k]["calls"] * kern[k]["avg_ms"]):
    rd, wr = 2 * fetch.get(k, 0) * 1024 / 1e9, write.get(k, 0) * 1024 / 1e9
    kern[k]["read_gb"], kern[k]["write_gb"] = rd, wr
    s = kern[k]
    lines.append(f"| {k} | {s['calls']} | {s['avg_ms']:.4f} | {s['full_launches']} | {s['full_avg_ms']:.4f} | {rd:.4f} | {wr:.4f} | "
                 f"{(rd + wr) / (s['full_avg_ms'] / 1e3):.0f} |")
if bench:
    lines += ["", "bench.py line of the traced run:", "", "```json", json.dumps(bench), "```"]
open(os.path.join(out, f"{tag}_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
source = f"profiles/{tag}_kernel_stats.csv + {tag}_pmc_summary.md (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE / WRITE_SIZE in separate passes)"
json.dump({"source": source, "kernels": kern}, open(os.path.join(out, "kernel_durations.json"), "w"), indent=1)
if bench and "vf::k_linearize_imu" in fetch:
    n_imu = bench["config"]["factors_per_gpu"]["imu"]
    k = "vf::k_linearize_imu"
    json.dump({"k1_bytes_per_imu_factor": (2 * fetch[k] + write[k]) * 1024 / n_imu, "read_bytes_per_imu_factor": 2 * fetch[k] * 1024 / n_imu,
               "write_bytes_per_imu_factor": write[k] * 1024 / n_imu,
               "source": f"profiles/{tag}_pmc_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                         "read = 2*FETCH_SIZE*1024 per MI355X_MICROARCH.md, write = WRITE_SIZE*1024)"},
              open(os.path.join(out, "traffic.json"), "w"), indent=1)
tl = subprocess.run([sys.executable, os.path.join(root, "tools", "step_timeline.py"), trace], capture_output=True, text=True).stdout
open(os.path.join(out, f"{tag}_step_timeline.txt"), "w").write(tl)
print("\n".join(lines[:14]))
print(tl[-400:])
