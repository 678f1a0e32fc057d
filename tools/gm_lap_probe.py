"""Host-side lap times of vf_solve (VF_SOLVE_TIMING, read once at the first call) on a steady fixed-lag GraphManager:
mean of each lap over the last solves.  usage (GPU box): python tools/gm_lap_probe.py [lag] [keyframes] [early_exit 0/1] 2> laps.txt"""
import os, sys, time
import numpy as np
os.environ["VF_SOLVE_TIMING"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if os.environ.get("VF_LIB"):                       # (A/B of two builds on one box)
    from vil_sensor_fusion_amd import _lib
    _lib._SO = os.path.abspath(os.environ["VF_LIB"])
from vil_sensor_fusion_amd import GraphManager, synth
lag, nkf = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 1300
paced = len(sys.argv) > 3 and sys.argv[3] == "paced"     # the device is idle when vf_solve is called, as at a 20-30 Hz keyframe rate (what it enqueued behind the previous solve has run)
seq = synth.make_sequence(seed=3, n_kf=nkf + 2)
gm = GraphManager(capacity=lag + 192, lag=lag, iterations=5)
gm.setInitialState(seq.gt_states[0])
gm.addIMUMeasurement(0.0, seq.imu_steps[0, 1:4], seq.imu_steps[0, 4:7])
t, times = 0.0, []
for k in range(1, nkf):
    for s in seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]:
        t += s[0]
        gm.addIMUMeasurement(t, s[1:4], s[4:7])
    gm.reserveNode(t)
    for i in np.nonzero(seq.btw_b == k)[0]:
        if seq.btw_a[i] >= 0:
            gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), np.eye(6) * seq.btw_cov[i])
    if paced:
        gm.lmStats()          # (synchronises the engine's stream)
    print(f"#solve {k}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    gm.solve()
    times.append((time.perf_counter() - t0) * 1e3)
print(("paced: " if paced else "back to back: ") + "vf_solve over the last 200 solves: mean %.3f ms, median %.3f, p99 %.3f" % (np.mean(times[-200:]), np.median(times[-200:]), np.percentile(times[-200:], 99)))
