"""fixed-lag vs whole-history handle on a dense stream of loop closures: how the gap depends on the lag and on the trials per solve"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_far_factors import _far_record
from tests.test_gpu_graph_manager import _stream
from vil_sensor_fusion_amd import synth
from vil_sensor_fusion_amd.graph_manager import GraphManager

n = 230
seq = synth.make_sequence(179, n)
traj_t, acc, gyr = _stream(seq)


def run(lag, iters, every, span_max=54):
    rng = np.random.default_rng(121)
    plan = {}
    k = 30
    while k < n - 5:
        span = int(rng.integers(8, min(span_max, k - 1)))
        plan[k] = (k - span, _far_record(seq, k - span, k, rng, cov=1e-3, noise=(3e-4, 3e-3)))
        k += int(rng.integers(every[0], every[1]))
    kw = dict(iterations=iters, rel_tol=0.0, abs_tol=0.0, max_far_factors=32)
    hs = {"lag": GraphManager(capacity=512, lag=lag, **kw), "whole": GraphManager(capacity=512, lag=0, **kw)}
    out = {k: [] for k in hs}
    for gm in hs.values():
        gm.setInitialState(seq.gt_states[0])
    i_imu, taken = 0, 0
    for k in range(1, n):
        for name, gm in hs.items():
            j = i_imu
            while j < traj_t.size and traj_t[j] <= seq.kf_time[k] + 0.01:
                gm.addIMUMeasurement(traj_t[j], acc[j], gyr[j]); j += 1
            gm.reserveNode(seq.kf_time[k])
            for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov):
                if b == k and a >= 1:
                    gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
        i_imu = j
        if k in plan and taken < 32:
            a, rec = plan[k]
            for gm in hs.values():
                gm.addBetweenFactor(a, k, (rec[0:4], rec[4:7]), np.eye(6) * 1e-3)
            taken += 1
        if taken >= 32 and k in plan:
            break
        for name, gm in hs.items():
            gm.solve()
            (q, t), v, b = gm.getState()
            out[name].append(np.concatenate([q, t, v, b]))
    for gm in hs.values():
        gm.close()
    a, b = np.array(out["lag"]), np.array(out["whole"])
    m = min(len(a), len(b))
    d = np.sqrt(np.sum((a[:m, 4:7] - b[:m, 4:7]) ** 2, axis=1))
    print(f"lag {lag:4d}, {iters:2d} trials per solve, a closure every {every[0]}..{every[1] - 1} keyframes ({taken} taken, {m} solves): fixed lag vs whole history "
          f"position rms {np.sqrt(np.mean(d ** 2)):.3e} m, max {d.max():.3e} m", flush=True)


for lag, iters, every in ((60, 5, (2, 6)), (60, 15, (2, 6)), (120, 5, (2, 6)), (60, 5, (8, 14)), (60, 5, (5, 9))):
    run(lag, iters, every)
