import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
seq = synth.make_sequence(0, 300)
eng = Engine(EngineOpts(windows=1, capacity=320))
for n in (1, 8, 64, 256):
    off = seq.imu_off[1:n + 2]
    eng.preintegrate(0, 1, off, seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV); eng.sync()
    t = []
    for _ in range(20):
        t0 = time.perf_counter(); eng.preintegrate(0, 1, off, seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV); eng.sync(); t.append(time.perf_counter() - t0)
    print(n, "factors:", f"{1e6*np.median(t):.1f} us per call; samples per factor", (off[1]-off[0]))
