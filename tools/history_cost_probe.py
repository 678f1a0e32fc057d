"""GPU probe: what one vf_solve costs as a whole-history handle (lag = 0, the reference's unbounded graph) grows -- LM with the
default termination rule and the reference-compat solve (one Gauss-Newton update) -- including the step at 1536 keyframes where
the refined solve switches itself on.  iSAM2 re-eliminates only the cliques a new factor touches; this library re-linearises and
re-solves the whole history (INTEGRATION.md "History length").  usage: python tools/history_cost_probe.py [n_max]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vil_sensor_fusion_amd import synth  # noqa: E402
from vil_sensor_fusion_amd.graph_manager import GraphManager  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8200
marks = [m for m in (250, 500, 1000, 1500, 1600, 2000, 3000, 4000, 6000, 8000) if m < n - 10]
seq = synth.make_sequence(seed=81, n_kf=n, keep_raw=True)
cov = {c: np.eye(6) * c for c in (synth.VIO_COV, synth.LIDAR_COV)}
by_end = {int(b): i for i, b in enumerate(seq.btw_b)}
for compat in (False, True):
    gm = GraphManager(capacity=4096, iterations=5, lag=0, reference_compat=compat)
    gm.setInitialState(seq.gt_states[0])
    gm.addIMUMeasurement(0.0, seq.imu_acc[0], seq.imu_gyro[0])
    i_imu, rows = 0, []
    for k in range(1, marks[-1] + 6):
        while i_imu < seq.imu_t.size and seq.imu_t[i_imu] <= seq.kf_time[k] + 0.011:
            gm.addIMUMeasurement(seq.imu_t[i_imu], seq.imu_acc[i_imu], seq.imu_gyro[i_imu])
            i_imu += 1
        gm.reserveNode(seq.kf_time[k])
        if k in by_end:
            i = by_end[k]
            if seq.btw_a[i] >= 1:
                gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), cov[float(seq.btw_cov[i])])
        timed = any(m <= k < m + 5 for m in marks)
        if timed or k % 10 == 0:
            t0 = time.perf_counter()
            gm.solve()
            dt = time.perf_counter() - t0
            if timed:
                rows.append((k, dt))
    print(f"## lag = 0, reference_compat = {compat} ({'one Gauss-Newton update per solve' if compat else 'LM, termination rule 1e-5 / 1e-5, at most 5 trials'})")
    for m in marks:
        ts = [dt for k, dt in rows if m <= k < m + 5]
        info = gm.solverInfo()
        print(f"history {m:5d} keyframes: vf_solve {np.mean(ts) * 1e3:7.2f} ms (min {np.min(ts) * 1e3:.2f})")
    print(f"   final: window {gm.solverInfo()[0]} keyframes, refinement corrections per solve {gm.solverInfo()[1]}, lm {gm.lmStats()}", flush=True)
    gm.close()
