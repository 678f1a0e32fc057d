"""The random-loop-closure stress of tests/test_gpu_far_factors.py on one handle, stopping at the first failed solve and
saying what the engine held.  usage (GPU box): python tools/far_stress_probe.py [lag] [capacity]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_gpu_far_factors import _far_record
from tests.test_gpu_graph_manager import _stream
from vil_sensor_fusion_amd import synth, VilFusionError
from vil_sensor_fusion_amd.graph_manager import GraphManager
lag, cap = int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = 260
seq = synth.make_sequence(79, n)
traj_t, acc, gyr = _stream(seq)
rng = np.random.default_rng(21)
plan, k = {}, 30
while k < n - 5:
    span = int(rng.integers(8, min(50 - 6, k - 1)))
    plan[k] = (k - span, _far_record(seq, k - span, k, rng, cov=1e-3, noise=(3e-4, 3e-3)))
    k += int(rng.integers(3, 12))
gm = GraphManager(capacity=cap, iterations=5, lag=lag, rel_tol=0, abs_tol=0)
gm.setInitialState(seq.gt_states[0])
i_imu = 0
for k in range(1, n):
    while i_imu < traj_t.size and traj_t[i_imu] <= seq.kf_time[k] + 0.01:
        gm.addIMUMeasurement(traj_t[i_imu], acc[i_imu], gyr[i_imu]); i_imu += 1
    gm.reserveNode(seq.kf_time[k])
    for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov):
        if b == k and a >= 1:
            gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
    if k in plan:
        a, rec = plan[k]
        try:
            gm.addBetweenFactor(a, k, (rec[0:4], rec[4:7]), np.eye(6) * 1e-3)
            print(f"key {k}: closure ({a}, {k}) taken", flush=True)
        except VilFusionError as exc:
            print(f"key {k}: closure ({a}, {k}) refused: {exc}", flush=True)
    try:
        gm.solve()
    except VilFusionError as exc:
        print(f"key {k}: solve failed: {exc}; lm {gm.lmStats()}", flush=True)
        break
print("done at key", k, gm.lmStats())
