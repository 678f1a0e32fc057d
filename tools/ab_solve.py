"""A/B of library variants on one box: stage timings of K4 (and K3) on the bench workload shape.
usage: python tools/ab_solve.py <variant.so> ...   (each run in a child process)"""
import os, subprocess, sys
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].endswith('.so')):
    for so in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, os.path.abspath(so)])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vil_sensor_fusion_amd import _lib
_lib._SO = sys.argv[1]
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
B, N = 1024, 1000
seq = synth.make_sequence(0, N)
eng = Engine(EngineOpts(windows=B, capacity=N))
rec = synth.between_records(seq)
for w in range(B):
    eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
    eng.set_between(w, seq.btw_a, seq.btw_b, rec)
    eng.set_states(w, 0, seq.gt_states[:1]); eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
    eng.set_range(w, 0, 1); eng.predict(w, 1, N - 1); eng.set_range(w, 0, N)
eng.linearize(0); eng.decide(init=True); eng.assemble(); eng.sync()
r = [eng.time_stage('solve', 5) for _ in range(3)]
a = [eng.time_stage('assemble', 5) for _ in range(3)] + [eng.time_stage('assemble_idle', 5) for _ in range(2)]
k1 = [eng.time_stage('linearize_imu', 10) for _ in range(3)]
print(os.path.basename(sys.argv[1]), 'K1 ms', ' '.join(f'{x:.3f}' for x in k1), ' solve ms', ' '.join(f'{x:.3f}' for x in r), ' assemble ms (last two: idle)', ' '.join(f'{x:.3f}' for x in a))
