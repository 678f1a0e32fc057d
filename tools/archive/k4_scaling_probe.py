"""K4 (one-wave-per-window sweep) launch time against the number of windows per CU: 256 / 512 / 768 / 1024 windows = 1 / 2 / 3 / 4
waves per CU.  usage: python tools/k4_scaling_probe.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
N = 1000
seq = synth.make_sequence(0, N)
rec = synth.between_records(seq)
for B in (256, 512, 768, 1024):
    eng = Engine(EngineOpts(windows=B, capacity=N, chunks=1, sweep_two_sided_max=0))
    for w in range(B):
        eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(w, seq.btw_a, seq.btw_b, rec)
        eng.set_states(w, 0, seq.gt_states[:1]); eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1); eng.predict(w, 1, N - 1); eng.set_range(w, 0, N)
    eng.linearize(0); eng.decide(init=True); eng.assemble(); eng.sync()
    t = [eng.time_stage('solve', 5) for _ in range(3)]
    print(f"B {B:5d}  solve ms {' '.join(f'{x:.3f}' for x in t)}   GB/s of traffic {B * N * 14160 / (min(t) * 1e-3) / 1e9:.0f}")
    eng.close()
