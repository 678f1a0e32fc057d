"""K3 + K4 (assemble, then the split band solve) against the assembling forward sweep (vf_engine_opts.solve_assemble_min:
k_band_forward_asm forms the block rows of H from the J stream on the matrix cores, K3 is not launched): agreement on a small
ragged batch (one staged trial, then LM + marginalised slides), then stage times at B windows x 1000 poses.
usage (GPU box): python tools/asm_probe.py [B ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("VF_LIB"):
    from vil_sensor_fusion_amd import _lib
    _lib._SO = os.path.abspath(os.environ["VF_LIB"])
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS


def build(B, n, seqs, ragged=False, **opts):
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
        eng.set_range(w, 0, n - ((w % 7) if ragged else 0))
    return eng


n, B0 = 100, 9
seqs = [synth.make_sequence(seed=40 + i, n_kf=n + 8) for i in range(5)]
WAVES = int(os.environ.get("VF_ASM_WAVES", "1"))
a, b = build(B0, n, seqs, True, solve_split_min=1), build(B0, n, seqs, True, solve_split_min=1, solve_assemble_min=1, solve_assemble_waves=WAVES)
for e in (a, b):
    e.linearize(); e.decide(init=True); e.assemble(); e.solve(); e.sync()
for w in range(B0):
    m = n - (w % 7)
    da, db = a.read_delta(w, 0, m), b.read_delta(w, 0, m)
    pa, pb = a.read_panels(w, 0, m), b.read_panels(w, 0, m)
    print(f"window {w} ({m} rows): first trial, max |delta| {np.abs(da).max():.3e}, |difference| {np.abs(da - db).max():.3e}; panels "
          f"{np.abs(pa - pb).max() / np.abs(pa).max():.3e} (relative to the largest entry); first bad row",
          next((k for k in range(m) if np.abs(pa[k] - pb[k]).max() > 1e-6 * np.abs(pa).max()), None), flush=True)
for e in (a, b):
    e.iterate(12)
    e.slide(); e.iterate(5)
    e.slide(); e.iterate(5)
worst = max(np.abs(a.get_states(w, 2, n - 8) - b.get_states(w, 2, n - 8)).max() for w in range(B0))
print("two-kernel vs assembling sweep, 12 + 5 + 5 trials and two marginalised slides: max |state difference|", worst,
      "lm", [a.read_lm(w)["accepted"] for w in range(B0)], [b.read_lm(w)["accepted"] for w in range(B0)], flush=True)
a.close(); b.close()
n = 1000
seqs = [synth.make_sequence(seed=80 + i, n_kf=n + 8) for i in range(16)]
for B in [int(x) for x in sys.argv[1:]] or [1024]:
    for mode in ("fused K4", "split", "assembling", "assembling, two waves"):
        e = build(B, n, seqs, solve_split_min=0 if mode == "fused K4" else 1, solve_assemble_min=1 if mode.startswith("assembling") else 0,
                  solve_assemble_waves=2 if mode.endswith("two waves") else 1)
        e.iterate(3)
        ts = [e.time_stage("solve", 5) for _ in range(3)]
        ta = [e.time_stage("assemble", 5) for _ in range(2)]
        t0 = time.perf_counter(); e.iterate(5); e.sync(); it = (time.perf_counter() - t0) * 1e3
        print(f"B = {B} {mode}: solve {' '.join(f'{x:.3f}' for x in ts)} ms; assemble {min(ta):.3f} ms; iterate(5) {it:.1f} ms", flush=True)
        e.close()
