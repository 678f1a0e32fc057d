#!/usr/bin/env python3
"""Prints the kernel sequence of the last fixed-lag update step of a rocprofv3 --kernel-trace run
(between the last two k_slide dispatches): start offset, duration and the idle gap in front of
each dispatch.  usage: tools/step_timeline.py <..._kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
slides = [i for i, r in enumerate(rows) if "k_slide" in r["Kernel_Name"]]
a, b = slides[-2], slides[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("vf::", "")
    print(f"{(s - t0) / 1e6:9.3f} ms  {name:28s} {(e - s) / 1e3:9.1f} us   gap {(s - prev_end) / 1e3:7.1f} us")
    busy += e - s
    prev_end = max(prev_end, e)
span = int(rows[b]["Start_Timestamp"]) - t0
print(f"step span {span / 1e6:.3f} ms, kernels busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
