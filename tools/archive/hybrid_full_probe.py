"""The update of bench.py under a termination rule that never fires (tolerances 1e-300: the hybrid launch path with every
window active in all five trials), between two phases with the rule off -- for a kernel trace (tools/trace_tail.py,
tools/trace_context.py): is the two-wave sweep slower under the hybrid launch than in the headline's?  It was, by 21 %: the roles of
its two waves were fixed and their SIMD placement depends on the kernel that ran before.  Compare
tools/build_variant.sh roles0 -DVF_ASM2_ROLES=0 (fixed roles) with the product library (roles agreed per CU).
usage: python tools/hybrid_full_probe.py <lib.so> [tol]"""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vil_sensor_fusion_amd import _lib
_lib._SO = os.path.abspath(sys.argv[1])
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-300
import bench
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
args = argparse.Namespace(window=1000, windows=1024, steps=8, warmup=2, init_iterations=200, iterations=5, host_workers=0, no_convergence_exit=False, sequences=0)
updates = bench.updates_per_engine(args)
seqs = bench.make_sequences(args, 0, 64, args.window + updates + 1)
eng, feed = bench.make_engine(args, 0, args.windows, seqs, updates)
fed = [0]
def step():
    eng.ingest_tail(*feed[fed[0]]); fed[0] += 1
    eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True); eng.iterate(5)
for _ in range(3): step()
eng.set_convergence(tol, tol)
for _ in range(2): step()
eng.sync()
t0 = time.perf_counter()
for _ in range(4): step()
eng.sync()
print(os.path.basename(sys.argv[1]), "tolerance", tol, "step ms", (time.perf_counter() - t0) / 4 * 1e3, flush=True)
eng.set_convergence(0.0, 0.0)          # third phase: the rule off again (the hybrid's buffers stay allocated)
for _ in range(3): step()
eng.sync()
