import sys, os, numpy as np
sys.path.insert(0, '.')
from vil_sensor_fusion_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != 'base':
    _lib._SO = os.path.abspath(f'scratch/libvf_{sys.argv[1]}.so')
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
B, N = 1024, 1000
seq = synth.make_sequence(0, N)
eng = Engine(EngineOpts(windows=B, capacity=N))
rec = synth.between_records(seq)
for w in range(B):
    eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
    eng.set_states(w, 0, seq.gt_states[:1]); eng.set_range(w, 0, 1)
    if w == 0: eng.predict(w, 1, N - 1); st = eng.get_states(0, 0, N)
    else: eng.set_states(w, 0, st)
    eng.set_range(w, 0, N)
ts = [eng.time_stage('linearize_imu', 20) for _ in range(5)]
n = eng.counts()['imu']
print(sys.argv[1] if len(sys.argv) > 1 else 'base', 'K1 ms', [round(t, 4) for t in ts], 'best GB/s', round(n * 5496 / min(ts) / 1e6, 1), 'median', round(n * 5496 / sorted(ts)[2] / 1e6, 1))
