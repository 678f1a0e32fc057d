"""A/B of library variants on the bench's update under the LM termination rule (bench.py `with_convergence_exit`): ms per
update step of 1 024 windows x 1 000 poses.  usage: python tools/ab_conv_exit.py <variant.so> ... (each in a child process)"""
import os, subprocess, sys
if len(sys.argv) > 2:
    for so in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, os.path.abspath(so)])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, time
from vil_sensor_fusion_amd import _lib
_lib._SO = sys.argv[1]
import bench
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
args = argparse.Namespace(window=1000, windows=1024, steps=16, warmup=2, init_iterations=200, iterations=5, host_workers=0, no_convergence_exit=False, sequences=0)
updates = bench.updates_per_engine(args)
seqs = bench.make_sequences(args, 0, 64, args.window + updates + 1)
eng, feed = bench.make_engine(args, 0, args.windows, seqs, updates)
fed = [0]
def step():
    eng.ingest_tail(*feed[fed[0]]); fed[0] += 1
    eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True); eng.iterate(5)
for _ in range(3): step()
eng.set_convergence(1e-5, 1e-5)
for _ in range(2): step()
eng.sync()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(5): step()
    eng.sync(); ts.append((time.perf_counter() - t0) / 5 * 1e3)
print(os.path.basename(sys.argv[1]), 'termination rule on: step ms', ' '.join(f'{x:.2f}' for x in ts), flush=True)
