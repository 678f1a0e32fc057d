"""Mean of every vf_solve lap over the last N solves of a tools/gm_lap_probe.py stderr log.  usage: gm_lap_summary.py laps.txt [N]"""
import sys, collections
lines = open(sys.argv[1]).read().splitlines()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
starts = [i for i, l in enumerate(lines) if l.startswith("#solve")]
acc, order = collections.defaultdict(list), []
for l in lines[starts[-N]:]:
    if l.startswith("[vf_solve]"):
        parts = l.split()
        name, us = " ".join(parts[1:-2]), float(parts[-2])
        if name not in order: order.append(name)
        acc[name].append(us)
tot = 0.0
for name in order:
    m = sum(acc[name]) / N
    tot += m
    print(f"{name:18s} {m:8.1f} us per solve ({len(acc[name]) / N:.2f} laps per solve)")
print(f"{'sum':18s} {tot:8.1f} us")
