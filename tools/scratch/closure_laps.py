"""lap times of vf_solve (VF_SOLVE_TIMING) with one loop closure alive at a 1 000-keyframe lag"""
import os, sys
os.environ["VF_SOLVE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_far_factors import _far_record
from tests.test_gpu_graph_manager import _stream
from vil_sensor_fusion_amd import synth
from vil_sensor_fusion_amd.graph_manager import GraphManager

n = 1040
seq = synth.make_sequence(7, n)
traj_t, acc, gyr = _stream(seq)
rng = np.random.default_rng(3)
gm = GraphManager(capacity=2048, lag=1000, iterations=5)
gm.setInitialState(seq.gt_states[0])
i_imu = 0
for k in range(1, n):
    while i_imu < traj_t.size and traj_t[i_imu] <= seq.kf_time[k] + 0.01:
        gm.addIMUMeasurement(traj_t[i_imu], acc[i_imu], gyr[i_imu]); i_imu += 1
    gm.reserveNode(seq.kf_time[k])
    for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov):
        if b == k and a >= 1:
            gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
    if k == 1010:
        rec = _far_record(seq, 500, k, rng, cov=1e-2, noise=(1e-3, 1e-2))
        gm.addBetweenFactor(500, k, (rec[0:4], rec[4:7]), np.eye(6) * 1e-2)
    if k >= 1005:
        sys.stderr.write(f"==== solve {k}\n"); sys.stderr.flush()
    gm.solve()
gm.close()
