import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle
from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seq = synth.make_sequence(0, N)
prob = helpers.build_problem(oracle, seq, perturb=0.01)
eng = Engine(EngineOpts(windows=1, capacity=max(64,N)))
helpers.load_engine(eng, 0, prob)
eng.linearize(0); eng.assemble(); eng.solve()
H, g = eng.read_normal(0, 0, N); d = eng.read_delta(0, 0, N)
rc, do = oracle.band_solve(H, g, 1e-5)
np.set_printoptions(linewidth=220, precision=3)
err = np.abs(d - do) / (np.abs(do) + 1e-30)
print('btw a', prob['btw_a'][:12], 'b', prob['btw_b'][:12])
print('rel err per (kf,dof):'); print(err)
ae = np.abs(d - do).max(axis=1) / np.abs(do).max()
print('per-kf abs err / max|do|:'); print(ae)
