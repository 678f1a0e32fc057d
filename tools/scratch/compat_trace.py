import sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vil_sensor_fusion_amd import synth
from vil_sensor_fusion_amd.graph_manager import GraphManager
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1650
seq = synth.make_sequence(seed=81, n_kf=n + 4, keep_raw=True)
cov = {c: np.eye(6) * c for c in (synth.VIO_COV, synth.LIDAR_COV)}
by_end = {int(b): i for i, b in enumerate(seq.btw_b)}
gm = GraphManager(capacity=4096, iterations=5, lag=0, reference_compat=True)
gm.setInitialState(seq.gt_states[0])
gm.addIMUMeasurement(0.0, seq.imu_acc[0], seq.imu_gyro[0])
i_imu = 0
for k in range(1, n):
    while i_imu < seq.imu_t.size and seq.imu_t[i_imu] <= seq.kf_time[k] + 0.011:
        gm.addIMUMeasurement(seq.imu_t[i_imu], seq.imu_acc[i_imu], seq.imu_gyro[i_imu]); i_imu += 1
    gm.reserveNode(seq.kf_time[k])
    if k in by_end:
        i = by_end[k]
        if seq.btw_a[i] >= 1:
            gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), cov[float(seq.btw_cov[i])])
    if k % 10 == 0 or k >= n - 3:
        t0 = time.perf_counter(); gm.solve(); dt = time.perf_counter() - t0
        if k >= n - 3: print("solve", k, dt * 1e3, "ms")
