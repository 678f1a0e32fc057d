"""What a vf_solve costs with k loop closures alive: a GraphManager at a 1 000-keyframe lag (and a whole-history
one) is fed a keyframe + one loop closure per solve until 32 are alive; the time of each solve, by closures alive.

    python tools/far_closure_cost.py > profiles/r06_far_closure_cost.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.test_gpu_far_factors import _far_record          # noqa: E402
from tests.test_gpu_graph_manager import _stream            # noqa: E402
from vil_sensor_fusion_amd import synth                     # noqa: E402
from vil_sensor_fusion_amd.graph_manager import GraphManager  # noqa: E402


def run(lag, warm=1010, closures=32, after=20, spans=(100, 900)):
    n = warm + 2 * closures + after + 2
    seq = synth.make_sequence(7, n)
    traj_t, acc, gyr = _stream(seq)
    rng = np.random.default_rng(3)
    gm = GraphManager(capacity=2048, lag=lag, iterations=5, max_far_factors=32)
    gm.setInitialState(seq.gt_states[0])
    i_imu, alive, rows = 0, 0, []
    for k in range(1, n):
        while i_imu < traj_t.size and traj_t[i_imu] <= seq.kf_time[k] + 0.01:
            gm.addIMUMeasurement(traj_t[i_imu], acc[i_imu], gyr[i_imu]); i_imu += 1
        gm.reserveNode(seq.kf_time[k])
        for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov):
            if b == k and a >= 1:
                gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
        new = False
        if k >= warm and (k - warm) % 2 == 0 and alive < closures:
            a = k - int(rng.integers(spans[0], spans[1]))
            rec = _far_record(seq, a, k, rng, cov=1e-2, noise=(1e-3, 1e-2))
            gm.addBetweenFactor(a, k, (rec[0:4], rec[4:7]), np.eye(6) * 1e-2)
            alive += 1
            new = True
        t0 = time.perf_counter()
        gm.solve()
        dt = (time.perf_counter() - t0) * 1e3
        if k >= warm - 10:
            rows.append((k, alive, new, dt))
    st = gm.lmStats()
    gm.close()
    print(f"## GraphManager, lag {lag}, 5 trials per solve at most (default termination rule), one keyframe per solve; a loop closure (span {spans[0]}..{spans[1]}) with every second keyframe from {warm} on")
    print("# keyframe  closures alive  new closure  vf_solve ms")
    for k, alive, new, dt in rows:
        print(f"{k:9d}  {alive:14d}  {'yes' if new else '   '}          {dt:9.3f}")
    by = {}
    for k, alive, new, dt in rows:
        if not new:
            by.setdefault(alive, []).append(dt)
    print("# closures alive -> vf_solve ms (median of the solves at that count, the solve that takes a new closure aside): " + ", ".join(f"{a}: {np.median(v):.2f}" for a, v in sorted(by.items()) if a in (0, 1, 2, 4, 8, 9, 12, 16, 24, 32)))
    print(f"# lm {st}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["lag", "whole", "leaving"]
    if "lag" in which:
        run(1000)
    if "whole" in which:
        run(0)
    if "leaving" in which:
        # anchors 900 .. 995 keyframes back: they leave the 1 000-keyframe lag within 100 solves -- every closure becomes a far end of
        # the window's linear far factor (k_marginalize<2>: a joint marginalisation per solve, up to 192 rows x 235 columns)
        run(1000, after=110, spans=(900, 995))
