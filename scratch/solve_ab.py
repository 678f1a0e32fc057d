import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vil_sensor_fusion_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != 'base':
    _lib._SO = os.path.abspath(f'scratch/libvf_{sys.argv[1]}.so')
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
B, N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024, 1000
seq = synth.make_sequence(0, N + 8)
eng = Engine(EngineOpts(windows=B, capacity=N + 8))
rec = synth.between_records(seq)
for w in range(B):
    eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
    eng.set_between(w, seq.btw_a, seq.btw_b, rec)
    eng.set_states(w, 0, seq.gt_states[:1]); eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
    eng.set_range(w, 0, 1)
    if w == 0: eng.predict(w, 1, N - 1); st = eng.get_states(0, 0, N)
    else: eng.set_states(w, 0, st)
    eng.set_range(w, 0, N)
eng.iterate(2)
a = [eng.time_stage('solve', 5) for _ in range(3)]
print(sys.argv[1] if len(sys.argv) > 1 else 'base', 'B', B, 'solve ms', [round(x, 3) for x in a], eng.read_lm(0)['cost'])
