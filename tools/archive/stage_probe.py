import sys, numpy as np
sys.path.insert(0, '.')
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
print('solve ms', eng.time_stage('solve', 5), 'assemble ms', eng.time_stage('assemble', 5), 'k1', eng.time_stage('linearize_imu', 5), 'k2', eng.time_stage('linearize_between', 5))
