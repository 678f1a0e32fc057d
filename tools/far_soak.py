"""Soak of the far-factor bookkeeping: a fixed-lag GraphManager that compacts (small capacity) and a roomy one are fed the
same stream with a loop closure between random keys of the window every few keyframes -- thousands of solves, hundreds of
closures converted to linear rows, re-expressed, folded into the prior, slots compacted under them -- and must publish the
same states to the last bit, without a failed solve.
usage (GPU box): python tools/far_soak.py [keyframes] [lag] [max_far_factors, 0 = the default 8] [largest gap between closures, default 30]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if os.environ.get("VF_LIB"):
    from vil_sensor_fusion_amd import _lib
    _lib._SO = os.path.abspath(os.environ["VF_LIB"])
from tests.test_gpu_far_factors import _far_record
from tests.test_gpu_graph_manager import _stream
from vil_sensor_fusion_amd import synth, VilFusionError
from vil_sensor_fusion_amd.graph_manager import GraphManager
n, lag = int(sys.argv[1]) if len(sys.argv) > 1 else 3000, int(sys.argv[2]) if len(sys.argv) > 2 else 200
max_far = (int(sys.argv[3]) if len(sys.argv) > 3 else 0) or 8
gap = int(sys.argv[4]) if len(sys.argv) > 4 else 30
seq = synth.make_sequence(83, n)
traj_t, acc, gyr = _stream(seq)
rng = np.random.default_rng(31)
plan, k = {}, lag // 2
while k < n - 5:
    span = int(rng.integers(8, min(lag - 6, k - 1)))
    plan[k] = (k - span, _far_record(seq, k - span, k, rng, cov=1e-3, noise=(3e-4, 3e-3)))
    k += int(rng.integers(min(4, gap - 1), gap))
handles = {"small": GraphManager(capacity=lag + 64, iterations=5, lag=lag, max_far_factors=max_far),
           "roomy": GraphManager(capacity=4 * lag + 256, iterations=5, lag=lag, max_far_factors=max_far)}
alive_max, ends = 0, []
for gm in handles.values():
    gm.setInitialState(seq.gt_states[0])
i_imu, taken, refused, worst, t0 = 0, 0, 0, 0.0, time.time()
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
        ok = []
        for gm in handles.values():
            try:
                gm.addBetweenFactor(a, k, (rec[0:4], rec[4:7]), np.eye(6) * 1e-3); ok.append(True)
            except VilFusionError as exc:
                assert exc.code == -6, exc
                ok.append(False)
        assert ok[0] == ok[1], (k, ok)
        taken += ok[0]; refused += not ok[0]
        if ok[0]:
            ends.append(k)
    ends = [b for b in ends if b > k - lag + 3]
    alive_max = max(alive_max, len(ends))
    out = []
    for name, gm in handles.items():
        try:
            gm.solve()
        except VilFusionError as exc:
            print(f"solve {k}, handle {name}: {exc}; closures planned around here: {[(kk, plan[kk][0]) for kk in plan if k - lag <= kk <= k]}", flush=True)
            raise
        (q, t), v, b = gm.getState()
        out.append(np.concatenate([q, t, v, b]))
    worst = max(worst, float(np.abs(out[0] - out[1]).max()))
    if k % 500 == 0 or k == n - 1:
        print(f"solve {k}: {taken} closures taken, {refused} refused for capacity; largest difference small vs roomy handle so far {worst:.3e}; "
              f"lm small {handles['small'].lmStats()}; {time.time() - t0:.0f} s", flush=True)
st = [gm.lmStats() for gm in handles.values()]
print(f"far soak: {n - 1} solves at lag {lag}, handles made for {max_far} far factors, {taken} loop closures through their whole life ({refused} refused for capacity, up to {alive_max} alive at once), largest difference between the compacting and the roomy handle {worst:.3e}, "
      f"failed solves {st[0]['solve_failures']} / {st[1]['solve_failures']}")
