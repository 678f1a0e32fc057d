"""Probe of the optional LM termination: trials taken by converged windows, per K4 form."""
import sys, os, numpy as np, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
N = 300
seq = synth.make_sequence(0, N + 8); rec = synth.between_records(seq)
for B, chunks in ((1, 0), (1, 1), (300, 1)):
    eng = Engine(EngineOpts(windows=B, capacity=N + 8, chunks=chunks))
    for w in range(B):
        eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(w, seq.btw_a, seq.btw_b, rec)
        eng.set_states(w, 0, seq.gt_states[:1]); eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
    eng.predict(-1, 1, N - 1)
    for w in range(B): eng.set_range(w, 0, N)
    eng.iterate(5); eng.sync()
    a = eng.read_lm(0)
    eng.set_convergence(1e-5, 1e-5)
    eng.iterate(5); eng.sync()
    b = eng.read_lm(0)
    print('B', B, 'chunks', chunks, 'before', a, 'after', b)
    eng.close()
