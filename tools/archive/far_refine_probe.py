"""Windows of 300 / 1 700 / 4 000 keyframes with loop closures (far factors), started at the refined optimum WITHOUT the
closures: distance from that start after every Gauss-Newton update / every 3 LM trials, refined (far rows in the operator, Woodbury
as the preconditioner) and unrefined (normal equations + Woodbury).  usage (GPU box): python tools/far_refine_probe.py"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import oracle
from tests import helpers
from tests.test_gpu_far_factors import _far_record
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
oracle.build()
for n, closures in ((300, ((20, 250), (60, 200))), (1700, ((100, 1650), (400, 1200))), (4000, ((100, 3900), (400, 1200), (2000, 3500)))):
    seq = synth.make_sequence(seed=14, n_kf=n)
    prob = helpers.build_problem(oracle, seq)
    ref = helpers.oracle_window(oracle, prob)
    for _ in range(6):
        oracle.gn_step(ref, refine=12)
    prob = dict(prob, states=ref.states.copy())
    rng = np.random.default_rng(15)
    fa, fb = np.array([c[0] for c in closures], dtype=np.int32), np.array([c[1] for c in closures], dtype=np.int32)
    far = np.stack([_far_record(seq, a, b, rng, cov=1e-4, noise=(1e-4, 1e-3)) for a, b in closures])
    for name, opts, mode in (("refined GN", dict(refine_iterations=12), "gn"), ("unrefined GN", dict(refine_iterations=0, lm_excursion=0), "gn"),
                             ("refined LM", dict(refine_iterations=12), "lm"), ("unrefined LM", dict(refine_iterations=0, lm_excursion=0), "lm"),
                             ("refined GN, no far", dict(refine_iterations=12), "gn0")):
        eng = Engine(EngineOpts(windows=1, capacity=n, **opts))
        helpers.load_engine(eng, 0, prob)
        if mode != "gn0":
            eng.set_extra_between(0, fa, fb, far)
        hist = []
        for _ in range(4):
            if mode.startswith("gn"):
                eng.isam_step(0.0); x = eng.get_estimate(0, 0, n)
            else:
                eng.iterate(3); x = eng.get_states(0, 0, n)
            hist.append(helpers.ate(x, prob["states"])[0])
        print(n, name, "refine_count", eng.refine_count(), "distance from the start per step:", " ".join(f"{h:.3e}" for h in hist), eng.read_lm(0), flush=True)
        eng.close()
