"""Prints the per-kernel rows of a rocprofv3 --kernel-trace --stats --output-format csv directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(f'{r["Name"][:48]:50s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:10.2f} min {float(r["MinNs"])/1e3:10.2f} max {float(r["MaxNs"])/1e3:10.2f}')
