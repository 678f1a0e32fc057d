"""s_memtime stamps of k_sep_solve (diagnostic build: tools/build_stamps.sh): usage tools/sep_stamps_probe.py <chunks>."""
import sys, ctypes as C, numpy as np, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
from vil_sensor_fusion_amd import _lib
_lib._SO = os.path.join(ROOT, 'tools/libvilfusion_stamps.so')
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
P = int(sys.argv[1]); N = 1000
seq = synth.make_sequence(0, N)
eng = Engine(EngineOpts(windows=1, capacity=N, chunks=P))
rec = synth.between_records(seq)
eng.preintegrate(0, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
eng.set_between(0, seq.btw_a, seq.btw_b, rec)
eng.set_states(0, 0, seq.gt_states[:1]); eng.set_prior(0, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
eng.set_range(0, 0, 1); eng.predict(0, 1, N - 1); eng.set_range(0, 0, N)
eng.linearize(0); eng.decide(init=True); eng.assemble(); eng.sync()
print('P', P, 'solve ms', eng.time_stage('solve', 3))
st = (C.c_ulonglong * 16)()
_lib.lib().vf_debug_sep_stamps(st)
names = ['schur+barrier (prev step) / setup', 'p load + barrier', 'pivot loop', 'stores + barrier', 'last schur', 'back substitution']
m = P - 1
for i, nme in enumerate(names):
    print(f'{nme:36s} {st[i]:10d} ticks  {st[i]/m:9.1f} per separator')
print('total ticks', sum(st))
