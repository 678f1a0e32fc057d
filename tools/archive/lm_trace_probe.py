"""GPU probe: LM trial by trial on an n-keyframe window from IMU dead reckoning (costs, lambda, accepted / rejected / provisional, ATE vs
the oracle's refined optimum).  usage: python tools/lm_trace_probe.py <n> <seed> <lm_excursion> <trials>"""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import oracle
from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
oracle.build()
n = int(sys.argv[1]); seed = int(sys.argv[2]); W = int(sys.argv[3]); trials = int(sys.argv[4])
seq = synth.make_sequence(seed=seed, n_kf=n)
prob = helpers.build_problem(oracle, seq)
ref = helpers.oracle_window(oracle, prob)
for _ in range(7): oracle.gn_step(ref, refine=12)
eng = Engine(EngineOpts(windows=1, capacity=n, lm_excursion=W))
helpers.load_engine(eng, 0, prob)
eng.reset_lambda(); eng.linearize(0); eng.decide(True)
print('start cost', eng.read_lm(0)['cost'])
for it in range(trials):
    eng.assemble(); eng.solve(); eng.retract(); eng.linearize(1); eng.decide(False)
    lm = eng.read_lm(0); ex = eng.read_excursions(0)
    print(it, f"cost {lm['cost']:.6f} lam {lm['lam']:.1e} acc {lm['accepted']} rej {lm['rejected']} prov {ex} ATE {helpers.ate(eng.get_states(0,0,n), ref.states)[0]:.3e}", flush=True)
