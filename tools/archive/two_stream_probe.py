"""Does the update step run faster as TWO half-batches on two HIP streams, out of phase (one half in the latency-bound
banded solve while the other is in the bandwidth-bound linearisation / assembly), than as one full batch?
usage: python tools/two_stream_probe.py [windows] [steps]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
args = argparse.Namespace(init_iterations=200, window=1000, windows=W, steps=steps, warmup=2, iterations=5, host_workers=0, no_convergence_exit=True, sequences=0)
updates = 3 * (steps + 2) + 2
seqs = bench.make_sequences(args, 0, 64, args.window + updates + 1)     # 64 distinct sequences are enough for a timing probe


def step(e):
    e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
    e.iterate(5)


def timed(engines, lag):
    for _ in range(2):
        for e in engines: step(e)
    for e in engines: e.sync()
    if lag and len(engines) > 1:
        engines[1].solve()              # one banded solve ahead of the loop: the second stream starts ~one K4 late
    t0 = time.perf_counter()
    for _ in range(steps):
        for e in engines: step(e)
    for e in engines: e.sync()
    return (time.perf_counter() - t0) / steps * 1e3


one = bench.make_engine(args, 0, W, seqs, updates, all_resident=True)[0]
t1 = timed([one], False)
print(f"one engine, {W} windows: {t1:.3f} ms per step", flush=True)
del one
halves = [bench.make_engine(args, 0, W // 2, seqs[i::2], updates, all_resident=True)[0] for i in range(2)]
t2 = timed(halves, False)
print(f"two engines x {W // 2} windows, streams in step: {t2:.3f} ms per step", flush=True)
t3 = timed(halves, True)
print(f"two engines x {W // 2} windows, second stream one solve late: {t3:.3f} ms per step", flush=True)
del halves
quarters = [bench.make_engine(args, 0, W // 4, seqs[i::4], updates, all_resident=True)[0] for i in range(4)]
t4 = timed(quarters, False)
print(f"four engines x {W // 4} windows: {t4:.3f} ms per step", flush=True)
