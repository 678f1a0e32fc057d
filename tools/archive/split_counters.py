"""summarise a rocprofv3 --pmc CSV (counter_collection) of tools/split_probe.py: per kernel (largest grid only) the mean of every counter"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
big = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "k_band" not in k:
        continue
    g = int(r["Grid_Size"])
    big[k] = max(big.get(k, 0), g)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "k_band" not in k or int(r["Grid_Size"]) != big[k]:
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, "grid", big[k], {n: round(sum(v) / len(v), 1) for n, v in c.items()})
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "SQ_WAVE_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m and m["SQ_BUSY_CU_CYCLES"]:
        print("   resident waves per busy CU-cycle (SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES):", round(m["SQ_WAVE_CYCLES"] / m["SQ_BUSY_CU_CYCLES"], 2))
