import sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import oracle
from tests import helpers
from tests.test_gpu_far_factors import _far_record
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
oracle.build()
n = 1000
seq = synth.make_sequence(seed=14, n_kf=n)
prob = helpers.build_problem(oracle, seq)
eng = Engine(EngineOpts(windows=1, capacity=1192))
helpers.load_engine(eng, 0, prob)
eng.iterate(3); eng.sync()
rng = np.random.default_rng(15)
far = np.stack([_far_record(seq, 100, 900, rng, cov=1e-4, noise=(1e-4, 1e-3))])
for rep in range(3):
    t0 = time.perf_counter()
    eng.set_extra_between(0, np.array([100], dtype=np.int32), np.array([900], dtype=np.int32), far)
    t1 = time.perf_counter()
    eng.iterate(1); eng.sync()
    t2 = time.perf_counter()
    print(f"set_extra_between {1e3 * (t1 - t0):.3f} ms, iterate(1)+sync {1e3 * (t2 - t1):.3f} ms")
