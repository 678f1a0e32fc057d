"""K4 as one fused kernel (k_band_solve) against the split form (k_band_forward with the compact 20 KB trailing window at two
waves per SIMD + k_band_backward): bit identity on a small batch, stage time of `solve` at B windows x 1000 poses.
usage (GPU box): python tools/split_probe.py [B ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("VF_LIB"):
    from vil_sensor_fusion_amd import _lib
    _lib._SO = os.path.abspath(os.environ["VF_LIB"])
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS


def build(B, n, seqs, **opts):
    eng = Engine(EngineOpts(windows=B, capacity=n + 8, chunks=1, sweep_two_sided_max=0, **opts))
    recs = [synth.between_records(s) for s in seqs]
    for w in range(B):
        s = seqs[w % len(seqs)]
        eng.preintegrate(w, 1, s.imu_off[1:n + 1], s.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        m = s.btw_b < n
        eng.set_between(w, s.btw_a[m], s.btw_b[m], recs[w % len(seqs)][m])
        eng.set_states(w, 0, s.gt_states[0].reshape(1, 16))
        eng.set_prior(w, 0, synth.prior_record(s.gt_states[0], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
    eng.predict(-1, 1, n - 1)
    for w in range(B):
        eng.set_range(w, 0, n)
    return eng


n = 200
seqs = [synth.make_sequence(seed=40 + i, n_kf=n + 8) for i in range(5)]
a, b = build(70, n, seqs, solve_split_min=0), build(70, n, seqs, solve_split_min=1)
for e in (a, b):
    e.iterate(12)
    e.slide(); e.iterate(5)
worst = max(np.abs(a.get_states(w, 1, n) - b.get_states(w, 1, n)).max() for w in range(70))
print("fused vs split, 70 windows x 200 poses, 12 + 5 trials and a marginalised slide: max |state difference|", worst,
      "lm equal:", all(a.read_lm(w) == b.read_lm(w) for w in range(70)), flush=True)
a.close(); b.close()
n = 1000
seqs = [synth.make_sequence(seed=80 + i, n_kf=n + 8) for i in range(16)]
for B in [int(x) for x in sys.argv[1:]] or [1024, 2048]:
    for split in (0, 1):
        e = build(B, n, seqs, solve_split_min=split)
        e.iterate(3)
        t = [e.time_stage("solve", 5) for _ in range(3)]
        t0 = time.perf_counter(); e.iterate(5); e.sync(); it = (time.perf_counter() - t0) * 1e3
        print(f"B = {B} {'split' if split else 'fused'}: solve {' '.join(f'{x:.3f}' for x in t)} ms = {min(t) * 1024 / B:.3f} ms per 1024 windows; iterate(5) {it:.1f} ms", flush=True)
        e.close()
