"""GPU probe: what one vf_solve costs as a whole-history handle (lag = 0, the reference's unbounded graph) grows -- LM with the
default termination rule, the reference-compat solve (one Gauss-Newton update over the whole history), and the same update done
incrementally (vf_graph_opts.incremental: the banded form of ISAM2::update -- only the keyframes from the first one that moved
are eliminated again).  One solve per keyframe, like the node.  usage: python tools/history_cost_probe.py [n_max] [modes]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vil_sensor_fusion_amd import synth  # noqa: E402
from vil_sensor_fusion_amd.graph_manager import GraphManager  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8200
want = sys.argv[2].split(",") if len(sys.argv) > 2 else ["lm", "compat", "incremental"]
all_marks = (250, 500, 1000, 1500, 1600, 2000, 3000, 4000, 6000, 8000, 12000, 16000, 20000)
seq = synth.make_sequence(seed=81, n_kf=n, keep_raw=True)
cov = {c: np.eye(6) * c for c in (synth.VIO_COV, synth.LIDAR_COV)}
by_end = {int(b): i for i, b in enumerate(seq.btw_b)}
MODES = {"lm": (dict(), 8200, "LM, termination rule 1e-5 / 1e-5, at most 5 trials"),
         "compat": (dict(reference_compat=True), 20200, "one Gauss-Newton update per solve over the whole history"),
         "incremental": (dict(reference_compat=True, incremental=True), 10 ** 9, "the same update, incremental (suffix re-elimination)")}
final = {}
for name in want:
    kw, cap, what = MODES[name]
    marks = [m for m in all_marks if m < min(n, cap) - 10]
    gm = GraphManager(capacity=4096, iterations=5, lag=0, **kw)
    gm.setInitialState(seq.gt_states[0])
    gm.addIMUMeasurement(0.0, seq.imu_acc[0], seq.imu_gyro[0])
    i_imu, rows, spans = 0, [], []
    every = 1 if name == "incremental" else 10          # (the whole-history solves are too slow to run 20 000 of them: one per 10 keyframes between marks)
    for k in range(1, marks[-1] + 6):
        while i_imu < seq.imu_t.size and seq.imu_t[i_imu] <= seq.kf_time[k] + 0.011:
            gm.addIMUMeasurement(seq.imu_t[i_imu], seq.imu_acc[i_imu], seq.imu_gyro[i_imu])
            i_imu += 1
        gm.reserveNode(seq.kf_time[k])
        if k in by_end:
            i = by_end[k]
            if seq.btw_a[i] >= 1:
                gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), cov[float(seq.btw_cov[i])])
        timed = any(m - 20 <= k < m + 5 for m in marks)
        if timed or k % every == 0:
            t0 = time.perf_counter()
            gm.solve()
            dt = time.perf_counter() - t0
            if any(m <= k < m + 5 for m in marks):
                rows.append((k, dt))
            if name == "incremental":
                info = gm.incrementalInfo()
                spans.append((k, k + 1 - info["first_eliminated_key"], k + 1 - info["last_substituted_key"], dt))
    print(f"## lag = 0, {name}: {what}")
    for m in marks:
        ts = [dt for k, dt in rows if m <= k < m + 5]
        extra = ""
        if spans:
            near = [s for s in spans if m - 500 <= s[0] < m + 5]
            extra = (f"   [last 500 updates: keyframes eliminated again median {np.median([s[1] for s in near]):.0f} p99 {np.percentile([s[1] for s in near], 99):.0f} max {max(s[1] for s in near)}; "
                     f"substituted again median {np.median([s[2] for s in near]):.0f} max {max(s[2] for s in near)}; vf_solve mean {np.mean([s[3] for s in near]) * 1e3:.2f} p99 {np.percentile([s[3] for s in near], 99) * 1e3:.2f} max {max(s[3] for s in near) * 1e3:.2f} ms]")
        print(f"history {m:5d} keyframes: vf_solve {np.mean(ts) * 1e3:7.2f} ms (min {np.min(ts) * 1e3:.2f}){extra}")
    k_end = marks[-1] + 4
    final[name] = gm.trajectory(0, k_end + 1)
    extra = f", incremental {gm.incrementalInfo()}" if name == "incremental" else ""
    print(f"   final: window {gm.solverInfo()[0]} keyframes, refinement corrections per solve {gm.solverInfo()[1]}, lm {gm.lmStats()}{extra}", flush=True)
    gm.close()
if "compat" in final and "incremental" in final:
    m = min(final["compat"].shape[0], final["incremental"].shape[0])
    d = final["compat"][:m, 4:7] - final["incremental"][:m, 4:7]
    print(f"## smoothed trajectory, incremental vs whole-history re-elimination (first {m} keyframes; NOT the same stream of solves: the whole-history "
          f"handle solved every 10th keyframe between marks): ATE {np.sqrt(np.mean(np.sum(d * d, axis=1))):.3e} m, last keyframe {np.linalg.norm(d[-1]):.3e} m")
for nm, tr in final.items():
    d = tr[:, 4:7] - seq.gt_states[:tr.shape[0], 4:7]
    print(f"## {nm}: ATE to the synthetic ground truth {np.sqrt(np.mean(np.sum(d * d, axis=1))):.4f} m over {tr.shape[0]} keyframes")
