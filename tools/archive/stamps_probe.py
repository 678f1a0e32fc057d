import sys, ctypes as C, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vil_sensor_fusion_amd import _lib
_lib._SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvilfusion_stamps.so')
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
B = int(sys.argv[1]); N = 1000
seq = synth.make_sequence(0, N)
eng = Engine(EngineOpts(windows=B, capacity=N))
rec = synth.between_records(seq)
for w in range(B):
    eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
    eng.set_between(w, seq.btw_a, seq.btw_b, rec)
    eng.set_states(w, 0, seq.gt_states[:1]); eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
    eng.set_range(w, 0, 1); eng.predict(w, 1, N - 1); eng.set_range(w, 0, N)
eng.linearize(0); eng.decide(init=True); eng.assemble(); eng.sync()
print('B', B, 'solve ms', eng.time_stage('solve', 3))
st = (C.c_ulonglong * 16)()
_lib.lib().vf_debug_solve_stamps(st)
names = ['loop top', 'fetch issue+panel LDS load', 'pivot chain', 'P write + panel store', 'schur mfma+wb', 'commit_row', 'bs: between steps', 'bs prep: rows->LDS (+ panel wait)', 'bs prep: column data + far couplings', 'bs: (prep -> solve)', 'bs solve: register chain + store']
# (the stamps carry scheduling barriers: the prep of step k-1 and the solve of step k overlap in the shipped build)
tot = sum(st)
for i, nme in enumerate(names):
    print(f'{nme:28s} {st[i]/N:9.1f} cycles/step  {100*st[i]/tot:5.1f}%')
print('total cycles/step', tot / N)
