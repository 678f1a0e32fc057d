"""far_batch_columns on / off with 1..8 far factors on one 70-keyframe window: do the two agree bit for bit?
usage (GPU box): python tools/far_batch_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import oracle
from tests import helpers
from tests.test_gpu_far_factors import _far_record
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
oracle.build()
total, n = 120, 70
seq = synth.make_sequence(seed=97, n_kf=total)
prob = helpers.build_problem(oracle, seq, perturb=0.002)
rng = np.random.default_rng(13)
allc = ((3, 60), (8, 41), (8, 66), (10, 50), (12, 64), (15, 58), (20, 69), (25, 45))
rec = np.stack([_far_record(seq, a, b, rng, cov=1e-3) for a, b in allc])
for m in (8,):
    fa, fb = np.array([c[0] for c in allc[:m]], dtype=np.int32), np.array([c[1] for c in allc[:m]], dtype=np.int32)
    res = []
    for batch in (1, 1, 0, 0):
        eng = Engine(EngineOpts(windows=1, capacity=total, far_batch_columns=batch))
        helpers.load_engine(eng, 0, prob, 0, n)
        eng.set_extra_between(0, fa, fb, rec[:m])
        d1 = None
        hist = []
        eng.iterate(12); hist.append(eng.get_states(0, 0, n))
        for sl in range(1, 16):
            eng.slide(marginalize=True); eng.iterate(4); hist.append(eng.get_states(0, sl, n))
        res.append((d1, hist, eng.read_lm(0)))
        eng.close()
    for i, j, name in ((0, 1, "batch vs batch"), (2, 3, "sequential vs sequential"), (0, 2, "batch vs sequential")):
        print(name, "per trial:", " ".join(f"{np.abs(a - b).max():.1e}" for a, b in zip(res[i][1], res[j][1])), res[i][2], flush=True)
