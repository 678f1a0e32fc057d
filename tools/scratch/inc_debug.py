import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle
from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
oracle.build()
SWEEP = dict(chunks=1, sweep_two_sided_max=0, solve_assemble_min=0, refine_iterations=0, lm_excursion=0)
n0, n = 40, 330
seq = synth.make_sequence(seed=91, n_kf=n)
prob = helpers.build_problem(oracle, seq, perturb=0.0)
inc = Engine(EngineOpts(windows=1, capacity=n + 8, incremental=1, **SWEEP))
helpers.load_engine(inc, 0, prob, 0, n0)
hi = n0
prev = inc.get_states(0, 0, n)
for u in range(120):
    inc.isam_step(1e-4)
    th = inc.get_states(0, 0, n)
    moved = np.nonzero(np.abs(th[:hi] - prev[:hi]).max(axis=1) > 0)[0]
    d = inc.read_delta(0, 0, hi)
    info = inc.incremental_info(0)
    big = np.abs(d).max(axis=1)
    comp = np.abs(d).argmax(axis=1)
    print(f"u {u:3d} hi {hi:3d} from {info['first_eliminated']:3d} stop {info['last_substituted']:3d} relinearised {moved[:6]}..{moved[-3:] if len(moved) else ''} ({len(moved)}) max|delta| {big.max():.2e} at kf {big.argmax()} comp {comp[big.argmax()]}; #kf with |d|>5e-5: {(big > 5e-5).sum()}")
    prev = th
    hi += 2
    inc.set_range(0, 0, hi)
