import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vil_sensor_fusion_amd import synth
from vil_sensor_fusion_amd.graph_manager import GraphManager
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seq = synth.make_sequence(seed=81, n_kf=n + 10, keep_raw=True)
cov = {c: np.eye(6) * c for c in (synth.VIO_COV, synth.LIDAR_COV)}
by_end = {int(b): i for i, b in enumerate(seq.btw_b)}
def run(name, **kw):
    gm = GraphManager(capacity=4096, iterations=5, lag=0, reference_compat=True, **kw)
    gm.setInitialState(seq.gt_states[0])
    gm.addIMUMeasurement(0.0, seq.imu_acc[0], seq.imu_gyro[0])
    i_imu, spans, ts = 0, [], []
    snaps = {}
    for k in range(1, n):
        while i_imu < seq.imu_t.size and seq.imu_t[i_imu] <= seq.kf_time[k] + 0.011:
            gm.addIMUMeasurement(seq.imu_t[i_imu], seq.imu_acc[i_imu], seq.imu_gyro[i_imu]); i_imu += 1
        gm.reserveNode(seq.kf_time[k])
        if k in by_end:
            i = by_end[k]
            if seq.btw_a[i] >= 1:
                gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), cov[float(seq.btw_cov[i])])
        t0 = time.perf_counter(); gm.solve(); ts.append(time.perf_counter() - t0)
        if kw.get("incremental"):
            info = gm.incrementalInfo()
            spans.append((k + 1 - info["first_eliminated_key"], k + 1 - info["last_substituted_key"]))
        if k in (n - 2, n - 1):
            snaps[k] = gm.trajectory(0, k + 1)
    tr = snaps[n - 1]
    d = tr[:, 4:7] - seq.gt_states[:n, 4:7]
    # how far does one update move each keyframe's estimate?
    mv = np.linalg.norm(snaps[n - 1][:n - 1, 4:7] - snaps[n - 2][:, 4:7], axis=1)
    s = np.array(spans[-500:]) if spans else None
    print(f"{name:28s} vf_solve last-500 mean {np.mean(ts[-500:]) * 1e3:6.2f} ms; ATE to gt {np.sqrt(np.mean(np.sum(d * d, axis=1))):.4f} m, last kf err {np.linalg.norm(d[-1]):.4f}; one update moves keyframe 10: {mv[10]:.2e} m, mid: {mv[n // 2]:.2e}, n-100: {mv[n - 100]:.2e}, n-10: {mv[n - 10]:.2e}" +
          (f"; eliminated again median {np.median(s[:, 0]):.0f} p90 {np.percentile(s[:, 0], 90):.0f}, substituted median {np.median(s[:, 1]):.0f}" if s is not None else ""), flush=True)
    gm.close()
    return tr
ref = run("full (refined when long)")
for thr in (1e-4, 1e-3, 1e-2, 1e-1):
    tr = run(f"incremental relin {thr:g}", incremental=True, relin_threshold=thr)
    d = tr[:, 4:7] - ref[:, 4:7]
    print(f"      vs full: ATE {np.sqrt(np.mean(np.sum(d * d, axis=1))):.3e} m, last keyframe {np.linalg.norm(d[-1]):.3e} m")
tr = run("full relin 1e-2", relin_threshold=1e-2)
tr = run("incremental relin 1e-4 wildfire 1e-5", incremental=True, wildfire=1e-5)
