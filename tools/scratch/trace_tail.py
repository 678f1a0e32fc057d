import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -75:]
t0 = int(rows[0]["Start_Timestamp"]); prev = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("vf::", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} us  {name:40s} {(e - s) / 1e3:8.1f} us   gap {(s - prev) / 1e3:7.1f} us")
    prev = max(prev, e)
