import os, sys, numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd())
sys.argv = ["x"]
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
n = 1000
seqs = [synth.make_sequence(seed=80 + i, n_kf=n + 8) for i in range(8)]
recs = [synth.between_records(s) for s in seqs]
for B in (512, 768, 1024, 1280, 1536, 2048):
    for waves in (1, 2):
        eng = Engine(EngineOpts(windows=B, capacity=n + 8, chunks=1, sweep_two_sided_max=0, solve_assemble_min=1, solve_assemble_waves=waves))
        for w in range(B):
            s = seqs[w % 8]
            eng.preintegrate(w, 1, s.imu_off[1:n + 1], s.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
            m = s.btw_b < n
            eng.set_between(w, s.btw_a[m], s.btw_b[m], recs[w % 8][m])
            eng.set_states(w, 0, s.gt_states[0].reshape(1, 16))
            eng.set_prior(w, 0, synth.prior_record(s.gt_states[0], REFERENCE_PRIOR_SIGMAS))
            eng.set_range(w, 0, 1)
        eng.predict(-1, 1, n - 1)
        for w in range(B):
            eng.set_range(w, 0, n)
        eng.iterate(2)
        t = min(eng.time_stage("solve", 5) for _ in range(3))
        print(f"B = {B:5d}, {waves} wave(s) per window: solve {t:.3f} ms = {t * 1024 / B:.3f} per 1024 windows", flush=True)
        eng.close()
