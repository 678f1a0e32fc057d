"""GPU probe: the library's defaults on windows past the refinement threshold (vf_engine_opts.refine_min_keyframes = 1536), several
lengths and sequences, from IMU dead reckoning: LM (refined solves + excursions) and Gauss-Newton (vf_engine_isam_step) against
the oracle's refined optimum of the same factors.  usage: python tools/long_window_sweep.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from tests import helpers  # noqa: E402
from vil_sensor_fusion_amd import Engine, EngineOpts, synth  # noqa: E402

oracle.build()
sizes = [int(a) for a in sys.argv[1:]] or [1600, 2500, 4000, 6000]
for n in sizes:
    for seed in (11, 12):
        seq = synth.make_sequence(seed=seed, n_kf=n)
        prob = helpers.build_problem(oracle, seq)
        ref = helpers.oracle_window(oracle, prob)
        t0 = time.time()
        for _ in range(7):
            oracle.gn_step(ref, refine=12)
        t_or = time.time() - t0
        out = []
        for mode in ("lm", "gn"):
            eng = Engine(EngineOpts(windows=1, capacity=n))
            helpers.load_engine(eng, 0, prob)
            hist = []
            if mode == "lm":
                for k in range(4):
                    eng.iterate(5)
                    hist.append(helpers.ate(eng.get_states(0, 0, n), ref.states)[0])
                lm, ex = eng.read_lm(0), eng.read_excursions(0)
                out.append(f"LM after 5/10/15/20 trials: {' '.join(f'{a:.1e}' for a in hist)} m (accepted {lm['accepted']} rejected {lm['rejected']} provisional {ex[0]} failed {lm['solve_failures']})")
            else:
                for k in range(6):
                    eng.isam_step(0.0)
                    hist.append(helpers.ate(eng.get_estimate(0, 0, n), ref.states)[0])
                out.append(f"GN per update: {' '.join(f'{a:.1e}' for a in hist)} m")
            eng.close()
        print(f"n = {n}, seed {seed} (oracle 7 refined GN updates {t_or:.1f} s): " + "; ".join(out), flush=True)
