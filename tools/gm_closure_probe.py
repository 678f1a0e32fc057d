"""A fixed-lag GraphManager (lag 1000) with one loop closure alive: the last solves, for tools/gm_timeline.py under rocprofv3
--kernel-trace.  usage (GPU box): python tools/gm_closure_probe.py [nkf]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vil_sensor_fusion_amd import GraphManager, synth
lag, nkf = 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 1020
a0, b0 = 40, 900
seq = synth.make_sequence(seed=3, n_kf=nkf + 2)
gm = GraphManager(capacity=lag + 192, lag=lag, iterations=5)
gm.setInitialState(seq.gt_states[0])
gm.addIMUMeasurement(0.0, seq.imu_steps[0, 1:4], seq.imu_steps[0, 4:7])
t = 0.0
for k in range(1, nkf):
    for s in seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]:
        t += s[0]
        gm.addIMUMeasurement(t, s[1:4], s[4:7])
    gm.reserveNode(t)
    for i in np.nonzero(seq.btw_b == k)[0]:
        if seq.btw_a[i] >= 0:
            gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), np.eye(6) * seq.btw_cov[i])
    if k == b0:
        Ra, Rb = synth.quat_to_rot(seq.gt_states[a0, :4]), synth.quat_to_rot(seq.gt_states[b0, :4])
        gm.addBetweenFactor(a0, b0, (synth.rot_to_quat(Ra.T @ Rb), Ra.T @ (seq.gt_states[b0, 4:7] - seq.gt_states[a0, 4:7])), np.eye(6) * 1e-4)
    st0 = gm.lmStats()
    t0 = time.perf_counter()
    gm.solve()
    if k >= nkf - 4 or b0 - 1 <= k <= b0 + 1:
        st1 = gm.lmStats()
        print("solve", k, f"{(time.perf_counter() - t0) * 1e3:.3f} ms, trials", st1["accepted"] + st1["rejected"] - st0["accepted"] - st0["rejected"], flush=True)
