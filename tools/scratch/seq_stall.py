"""does a long run of sequential Woodbury columns (no column engine) stall?  one window, 32 far factors, far_batch_columns = 0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle
from tests import helpers
from tests.test_gpu_far_factors import _far_record
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
oracle.build()
n = 1000
seq = synth.make_sequence(seed=5, n_kf=n)
prob = helpers.build_problem(oracle, seq)
rng = np.random.default_rng(1)
pairs = sorted({(int(a), int(a) + int(s)) for a, s in zip(rng.integers(2, 500, 40), rng.integers(50, 400, 40))})[:32]
fa, fb = np.array([a for a, _ in pairs], dtype=np.int32), np.array([b for _, b in pairs], dtype=np.int32)
far = np.array([_far_record(seq, a, b, rng, cov=1e-2, noise=(1e-3, 1e-2)) for a, b in pairs])
eng = Engine(EngineOpts(windows=1, capacity=n, max_far_factors=32, far_batch_columns=int(sys.argv[1]) if len(sys.argv) > 1 else 0))
helpers.load_engine(eng, 0, prob)
eng.set_extra_between(0, fa, fb, far)
ts = []
for i in range(60):
    t0 = time.perf_counter()
    eng.iterate(5)
    eng.read_lm(0)
    ts.append((time.perf_counter() - t0) * 1e3)
print("iterate(5) ms:", " ".join(f"{t:.0f}" for t in ts))
print("lm", eng.read_lm(0))
