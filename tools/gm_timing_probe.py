import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vil_sensor_fusion_amd import GraphManager, synth
lag, nkf = int(sys.argv[1]), int(sys.argv[2])
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
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
    if k == nkf - 3:
        os.environ["VF_SOLVE_TIMING"] = "1"
    st0 = gm.lmStats()
    t0 = time.perf_counter()
    gm.solve()
    st1 = gm.lmStats()
    if k >= nkf - 6 or k % every == 0:
        print(k, "trials", st1["accepted"] - st0["accepted"], "+", st1["rejected"] - st0["rejected"], "cost", st0["cost"], "->", st1["cost"])
    if k >= nkf - 3:
        print("solve", k, (time.perf_counter() - t0) * 1e3, "ms")
