#!/usr/bin/env python3
"""Kernel sequence of the LAST vf_solve of a rocprofv3 --kernel-trace run of tools/gm_timing_probe.py (from the last
k_preintegrate_t dispatch on): start offset, duration, gap in front.  usage: tools/gm_timeline.py <..._kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_preintegrate" in r["Kernel_Name"]]
a = marks[-1]
margs = [i for i, r in enumerate(rows) if "k_marginalize" in r["Kernel_Name"] and a - 8 <= i < a]      # (asynchronous staging: it runs beside K0, on a second stream)
if margs:
    a = margs[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end, busy = t0, 0
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("vf::", "")
    print(f"{(s - t0) / 1e3:9.1f} us  {name:34s} {(e - s) / 1e3:8.1f} us   gap {(s - prev_end) / 1e3:7.1f} us")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(prev_end - t0 - busy) / 1e3:.1f} us")
