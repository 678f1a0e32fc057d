"""condense the rocprofv3 --pmc CSV of tools/k6_probe.py into a markdown table: per K6 kernel instantiation (largest grid
only) mean counters per launch, float64 VALU operations and the fraction of the vector float64 peak they amount to at the
launch duration given in a kernel-trace CSV.  usage: k6_counters.py <counter_collection.csv> <kernel_trace.csv> <out.md>"""
import csv, sys, collections, re
pmc, trace, out = sys.argv[1:4]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    k = r["Kernel_Name"]
    if "k_degeneracy" in k and int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) >= (1 << 22):
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(pmc)):
    k = r["Kernel_Name"]
    if "k_degeneracy" in k and int(r["Grid_Size"]) >= (1 << 22):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = {4: "e_opt", 19: "condition_number", 0: "d_opt"}


def parse(k):
    """('f64' | 'f32', metric id) from a demangled `k_degeneracy<double, 6, 4>(...)` or a mangled `...IdLi6ELi4EE...` name"""
    m = re.search(r"k_degeneracy<(double|float), (\d+), (\d+)>", k) or re.search(r"k_degeneracyI([df])Li(\d+)ELi(\d+)E", k)
    return ("f64" if m.group(1) in ("double", "d") else "f32", int(m.group(3))) if m else ("?", -1)
PEAK_F64, PEAK_F32 = 78.6e12, 157.3e12     # MI355X vector peaks (MI355X_MICROARCH.md), FMA = 2 flop
L = ["# K6 on 2^22 6x6 matrices: VALU counters per launch (rocprofv3 --pmc, separate pass) and the compute roofline", "",
     "| kernel | dtype | ms per launch (trace) | ns per matrix | VALU insts / wave | f64 FMA | f64 MUL | f64 ADD | f64 TRANS | GFLOP per launch | TFLOP/s | of vector peak | HBM GB/s (296 or 148 B per matrix) |",
     "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    dt, mid = parse(k)
    is64 = dt == "f64"
    tag = names.get(mid, k[:40])
    ms = min(dur[k]) if dur.get(k) else float("nan")
    waves = c.get("SQ_WAVES", 0.0)
    fma, mul, add, tr = (c.get("SQ_INSTS_VALU_FMA_F64", 0), c.get("SQ_INSTS_VALU_MUL_F64", 0), c.get("SQ_INSTS_VALU_ADD_F64", 0), c.get("SQ_INSTS_VALU_TRANS_F64", 0))
    flop = (2 * fma + mul + add + tr) * 64
    tf = flop / (ms * 1e-3) / 1e12 if ms == ms else float("nan")
    nbytes = (296 if is64 else 148) * (1 << 22)
    L.append(f"| {tag} | {'f64' if is64 else 'f32'} | {ms:.4f} | {ms * 1e6 / (1 << 22):.4f} | {c.get('SQ_INSTS_VALU', 0) / max(waves, 1):.0f} | {fma:.3e} | {mul:.3e} | {add:.3e} | {tr:.3e} | "
             f"{flop / 1e9:.2f} | {tf:.2f} | {tf * 1e12 / (PEAK_F64 if is64 else PEAK_F32):.3f} | {nbytes / (ms * 1e-3) / 1e9:.0f} |")
L += ["", "f64 counters count wave-level instructions (x 64 lanes = operations); FMA counts 2 flop.  The float32 rows carry no float64 "
      "instructions: their VALU count per wave is the comparable figure.  Peak: 78.6 TFLOP/s vector float64 (256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz)."]
open(out, "w").write("\n".join(L) + "\n")
print("\n".join(L))
