"""Latency of the drop-in surface: GraphManager.solve() per keyframe in fixed-lag mode (lag keyframes), fed like
the node: IMU at 200 Hz, camera keyframes at 20 Hz, LiDAR at 10 Hz (synthetic Carla-like data).
usage: tools/graph_manager_probe.py <lag> <keyframes>"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vil_sensor_fusion_amd import GraphManager, synth
lag, nkf = int(sys.argv[1]), int(sys.argv[2])
seq = synth.make_sequence(seed=3, n_kf=nkf + 2)
gm = GraphManager(capacity=lag + 192, lag=lag, iterations=5)
t_kf = seq.t_kf if hasattr(seq, 't_kf') else None
steps, off = seq.imu_steps, seq.imu_off
times = []
t = 0.0
keys = []
for k in range(1, nkf):
    for s in steps[off[k]:off[k + 1]]:
        t += s[0]
        gm.addIMUMeasurement(t, s[1:4], s[4:7])
    key = gm.reserveNode(t)
    keys.append(key)
    # between factor to the previous keyframe of this source (identity-ish measurement from ground truth)
    if len(keys) >= 2:
        a, b = k - 1, k
        rec = synth.between_records(seq)[0] if False else None
    t0 = time.perf_counter()
    gm.solve()
    times.append(time.perf_counter() - t0)
times = np.array(times) * 1e3
warm = times[lag + 20:] if len(times) > lag + 40 else times[len(times) // 2:]
print(f'lag {lag}: solve() ms  first-half mean {times[:len(times)//2].mean():.3f}  steady mean {warm.mean():.3f}  p50 {np.median(warm):.3f}  p99 {np.percentile(warm, 99):.3f}  (n={len(warm)})')
