"""Durations (us) of the last N dispatches of one kernel in a rocprofv3 kernel-trace csv."""
import csv, glob, sys
d, name, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(name, len(rows), [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[-n:]])
