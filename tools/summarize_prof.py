#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the tracked summaries under profiles/.

usage: tools/summarize_prof.py <round-tag> <prof-dir> [out-dir]     (prof-dir = gpurun_out/prof_<tag>, written by tools/profile_round.sh)

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
# optional third argument: where to write (tools/profile_round.sh summarises ON the GPU box into gpurun_out/.../summary, the
# raw per-dispatch counter CSVs are too large to travel back; the files are then copied into profiles/ by hand)
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles")
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


def pmc_multi(sub):
    """{kernel: {counter: mean over the kernel's FULL launches}} of a pass that collected several counters: a dispatch's
    counters are rows with the same Dispatch_Id; full launches = dispatches whose SQ_WAVE_CYCLES (or first counter) is at
    least half of the kernel's largest."""
    f = one(f"{sub}/**/*counter_collection.csv")
    if not f:
        return {}
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        disp[(short(r["Kernel_Name"]), r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    per = collections.defaultdict(list)
    for (k, _), c in disp.items():
        per[k].append(c)
    out = {}
    for k, rows in per.items():
        key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in rows[0] else sorted(rows[0])[0]
        top = max(r.get(key, 0.0) for r in rows)
        full = [r for r in rows if r.get(key, 0.0) >= 0.5 * top] or rows
        out[k] = {c: sum(r.get(c, 0.0) for r in full) / len(full) for c in full[0]}
        out[k]["_full_launches"] = len(full)
    return out


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
# ---- SQ counters (two / three passes of 8): MFMA utilisation, VALU issue occupancy, waits, LDS conflicts
sq = {}
for sub in ("sq_a", "sq_b", "sq_c"):
    for k, c in pmc_multi(sub).items():
        sq.setdefault(k, {}).update({n: v for n, v in c.items() if not (n == "SQ_WAVE_CYCLES" and sub != "sq_a" and "SQ_WAVE_CYCLES" in sq.get(k, {}))})
if sq:
    hot = [k for k in ("vf::k_linearize_imu", "vf::k_linearize_between_prior", "vf::k_assemble", "vf::k_band_solve", "vf::k_band_forward_asm", "vf::k_band_forward_asm2", "vf::k_band_forward",
                       "vf::k_band_backward", "vf::k_retract", "vf::k_decide") if k in sq]
    L = [f"# {tag}: SQ counters per kernel (rocprofv3 --pmc, separate passes; means over full launches of `bench.py --steps 3`)", "",
         "Units (MI355X_MICROARCH.md, `s_memtime` tick vs SQ PMC units): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; "
         "SQ_VALU_MFMA_BUSY_CYCLES counts cycles of a SIMD's matrix pipe; SQ_BUSY_CYCLES is per shader engine x its busy time; "
         "SQ_INSTS_VALU_MFMA_MOPS_F64 counts 512-flop units (16x16x4 f64 = 2048 flop = 4 units per wave instruction).", "",
         "Derived: wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked in s_waitcnt / barrier); issue-stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; "
         "valu = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (share of a wave's life spent issuing vector instructions incl. MFMA); "
         "mfma-pipe = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs) = utilisation of the chip's matrix pipes; "
         "lds-conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; f64 VALU flop = (2 FMA + MUL + ADD) x 64 lanes.", "",
         "| kernel | waves | wait | issue-stall | valu | lds | mfma-pipe util | MFMA f64 GFLOP/launch | VALU f64 GFLOP/launch | VALU insts/wave | LDS insts/wave | lds-conflict | VMEM rd / wr per wave |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    summ = {}
    for k in hot:
        c = sq[k]
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        waves = c.get("SQ_WAVES", 0.0) or 1.0
        dur_ms = kern.get(k, {}).get("full_avg_ms", 0.0)
        pipe = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (dur_ms * 1e-3 * 2.4e9 * 1024) if dur_ms else 0.0
        mfma_gf = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512 / 1e9
        valu_gf = (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0)) * 64 / 1e9
        d = {"waves": waves, "wait_frac": c.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
             "valu_issue_frac": c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, "lds_issue_frac": c.get("SQ_ACTIVE_INST_LDS", 0.0) / wc,
             "mfma_pipe_util": pipe, "mfma_f64_gflop_per_launch": mfma_gf, "valu_f64_gflop_per_launch": valu_gf,
             "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0.0) / waves, "lds_insts_per_wave": c.get("SQ_INSTS_LDS", 0.0) / waves,
             "lds_conflict_frac": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / (c.get("SQ_LDS_IDX_ACTIVE", 0.0) or 1.0),
             "vmem_rd_per_wave": c.get("SQ_INSTS_VMEM_RD", 0.0) / waves, "vmem_wr_per_wave": c.get("SQ_INSTS_VMEM_WR", 0.0) / waves,
             # wave-level instruction counts the issue view of bench.py's roofline_solve is built from
             "valu_insts_per_launch": c.get("SQ_INSTS_VALU", 0.0),
             "valu_f64_insts_per_launch": c.get("SQ_INSTS_VALU_FMA_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) + c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0),
             "mfma_f64_insts_per_launch": c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / 4.0,
             "mfma_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
             "raw": {n: v for n, v in c.items()}}
        summ[k] = d
        if k in kern:
            kern[k]["sq"] = {n: v for n, v in d.items() if n != "raw"}
        L.append(f"| {k} | {waves:.0f} | {d['wait_frac']:.2f} | {d['issue_stall_frac']:.2f} | {d['valu_issue_frac']:.2f} | {d['lds_issue_frac']:.2f} | {pipe:.3f} | "
                 f"{mfma_gf:.2f} | {valu_gf:.2f} | {d['valu_insts_per_wave']:.0f} | {d['lds_insts_per_wave']:.0f} | {d['lds_conflict_frac']:.3f} | {d['vmem_rd_per_wave']:.0f} / {d['vmem_wr_per_wave']:.0f} |")
    L += ["", "Raw counter means per full launch:", "", "```json", json.dumps({k: summ[k]["raw"] for k in summ}, indent=1), "```"]
    open(os.path.join(out, f"{tag}_sq_counters.md"), "w").write("\n".join(L) + "\n")
    json.dump({"source": source, "kernels": kern}, open(os.path.join(out, "kernel_durations.json"), "w"), indent=1)
    print("\n".join(L[6:6 + 2 + len(hot)]))
tl = subprocess.run([sys.executable, os.path.join(root, "tools", "step_timeline.py"), trace], capture_output=True, text=True).stdout
open(os.path.join(out, f"{tag}_step_timeline.txt"), "w").write(tl)
print("\n".join(lines[:14]))
print(tl[-400:])
