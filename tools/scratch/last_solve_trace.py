"""print the kernel timeline of the last vf_solve in a rocprofv3 --kernel-trace csv (argument: the csv)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_read_result" in r["Kernel_Name"]]
i0, i1 = idx[-3] + 1, idx[-1] + 1
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:i1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vf::", "")[:44]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n}")
    prev_end = e
