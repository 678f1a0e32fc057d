"""GPU probe: how many corrections the refined solve needs as a function of the window length: res . M^-1 res after each correction,
relative to its first value, for Gauss-Newton updates from dead reckoning.  usage: python tools/refine_trace_sizes.py [n ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from tests import helpers  # noqa: E402
from vil_sensor_fusion_amd import Engine, EngineOpts, synth  # noqa: E402

oracle.build()
for n in [int(a) for a in sys.argv[1:]] or [1600, 2500, 4000, 6000]:
    seq = synth.make_sequence(seed=11, n_kf=n)
    prob = helpers.build_problem(oracle, seq)
    eng = Engine(EngineOpts(windows=1, capacity=n, refine_iterations=14, chunks=max(2, int(np.sqrt(n)) // 4 * 4)))
    helpers.load_engine(eng, 0, prob)
    for it in range(3):
        eng.gn_begin(0.0)
        eng.assemble(), eng.solve_local(), eng.solve_global()
        eng.refine_begin()
        row = []
        for c in range(14):
            eng.solve_local(), eng.solve_global(), eng.refine_step()
            k, red = eng.read_refine(0)
            row.append(f"{red:.0e}" + ("" if k == c + 1 else "*"))
        eng.refine_end()
        eng.retract()
        print(f"n = {n}, GN update {it}: " + " ".join(row), flush=True)
    eng.close()
