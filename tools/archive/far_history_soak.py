"""Whole-history handles (lag = 0, the reference's mode) with loop closures, across growth and across the refinement threshold
(1 536 keyframes): one that starts with 256 slots and grows, one with room from the start; default termination rule.  The
chunk geometry follows the capacity, so the two agree to rounding, not to the bit.
usage (GPU box): python tools/far_history_soak.py [keyframes]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_gpu_far_factors import _far_record
from tests.test_gpu_graph_manager import _stream
from vil_sensor_fusion_amd import synth, VilFusionError
from vil_sensor_fusion_amd.graph_manager import GraphManager
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2300
seq = synth.make_sequence(85, n)
traj_t, acc, gyr = _stream(seq)
rng = np.random.default_rng(33)
ends = sorted(int(x) for x in rng.choice(np.arange(200, n - 50), size=8, replace=False))
plan = {k: (int(rng.integers(5, k - 20)), None) for k in ends}
plan = {k: (a, _far_record(seq, a, k, rng, cov=1e-3, noise=(3e-4, 3e-3))) for k, (a, _) in plan.items()}
handles = {"grows": GraphManager(capacity=256, iterations=5, lag=0), "roomy": GraphManager(capacity=4096, iterations=5, lag=0)}
for gm in handles.values():
    gm.setInitialState(seq.gt_states[0])
i_imu, worst, t0, times = 0, 0.0, time.time(), []
for k in range(1, n):
    j = i_imu
    for gm in handles.values():
        j = i_imu
        while j < traj_t.size and traj_t[j] <= seq.kf_time[k] + 0.01:
            gm.addIMUMeasurement(traj_t[j], acc[j], gyr[j]); j += 1
        gm.reserveNode(seq.kf_time[k])
        for a, b, q, t, c in zip(seq.btw_a[seq.btw_b == k], seq.btw_b[seq.btw_b == k], seq.btw_q[seq.btw_b == k], seq.btw_t[seq.btw_b == k], seq.btw_cov[seq.btw_b == k]):
            if a >= 1:
                gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
    i_imu = j
    if k in plan:
        a, rec = plan[k]
        for gm in handles.values():
            gm.addBetweenFactor(a, k, (rec[0:4], rec[4:7]), np.eye(6) * 1e-3)
        print(f"key {k}: closure ({a}, {k})", flush=True)
    if k % 5 and k != n - 1 and k not in plan:
        continue
    out = []
    for name, gm in handles.items():
        ts = time.perf_counter()
        gm.solve()
        if name == "roomy":
            times.append((k, (time.perf_counter() - ts) * 1e3))
        (q, t), v, b = gm.getState()
        out.append(np.concatenate([q, t, v, b]))
    d = float(np.abs(out[0] - out[1]).max())
    worst = max(worst, d)
    if k % 250 == 0 or k == n - 1:
        recent = [x for kk, x in times if kk > k - 250]
        print(f"solve at key {k}: growing vs roomy handle {d:.3e} (worst so far {worst:.3e}); solver info {handles['grows'].solverInfo()}; lm {handles['grows'].lmStats()}; "
              f"vf_solve (roomy) mean over the last 250 keys {np.mean(recent):.2f} ms; {time.time() - t0:.0f} s", flush=True)
st = [gm.lmStats() for gm in handles.values()]
print(f"whole-history soak: {n - 1} keyframes, a solve every 5, 8 loop closures kept for good; worst difference growing vs roomy handle {worst:.3e}; failed solves {st[0]['solve_failures']} / {st[1]['solve_failures']}")
