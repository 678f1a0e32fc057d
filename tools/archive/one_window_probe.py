"""one 1000-pose window: K4p solve time and LM update latency.  usage: python tools/one_window_probe.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
N = 1000
seq = synth.make_sequence(0, N + 20)
eng = Engine(EngineOpts(windows=1, capacity=N + 20))
eng.preintegrate(0, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
eng.set_between(0, seq.btw_a, seq.btw_b, synth.between_records(seq))
eng.set_states(0, 0, seq.gt_states[:1]); eng.set_prior(0, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
eng.set_range(0, 0, 1); eng.predict(0, 1, N - 1); eng.set_range(0, 0, N)
eng.iterate(5); eng.sync()
print("one window: solve ms", " ".join(f"{eng.time_stage('solve', 10):.4f}" for _ in range(3)))
eng.iterate(1); eng.sync()
t = []
for _ in range(10):
    eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
    t0 = time.perf_counter(); eng.iterate(5); eng.sync(); t.append(time.perf_counter() - t0)
print("one window: iterate(5) ms", f"{1e3 * np.mean(t[2:]):.3f}")
