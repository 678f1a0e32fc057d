"""For every dispatch of one kernel in a rocprofv3 kernel-trace csv (the last N): its duration and the kernels dispatched
between it and the previous dispatch of the same kernel.  usage: trace_context.py <dir> <kernel> <N>"""
import csv, glob, sys
d, name, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
short = lambda r: r["Kernel_Name"].split("(")[0].replace("vf::", "").replace("void ", "")
for a, b in list(zip(idx[:-1], idx[1:]))[-n:]:
    dur = (int(rows[b]["End_Timestamp"]) - int(rows[b]["Start_Timestamp"])) / 1e3
    between = " ".join(f"{short(r)}:{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in rows[a + 1:b])
    print(f"{dur:7.0f} us  after: {between}")
