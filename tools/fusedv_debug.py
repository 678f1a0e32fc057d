"""Compare H, g of the lane-per-factor fused kernel (VF_FUSED=2) with K1 -> K3, block by block."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
n, lo, hi = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 0, 90
def mk(mode):
    os.environ["VF_FUSED"] = str(mode)
    eng = Engine(EngineOpts(windows=2, capacity=n, chunks=1))
    for w in range(2):
        seq = synth.make_sequence(seed=40 + w, n_kf=n)
        eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(w, seq.btw_a, seq.btw_b, synth.between_records(seq))
        eng.set_states(w, 0, seq.gt_states[:1])
        eng.set_prior(w, lo, synth.prior_record(seq.gt_states[lo], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1); eng.predict(w, 1, hi - 1)
        rng = np.random.default_rng(40 + w)
        st = eng.get_states(w, 0, hi); st[1:, 4:10] += rng.normal(size=(hi - 1, 6)) * 0.02
        eng.set_states(w, 0, st); eng.set_range(w, lo, hi)
    return eng
f, u = mk(2), mk(0)
u.linearize(0); u.decide(init=True); u.assemble(); u.sync()
f.linearize(0); f.decide(init=True)          # (between / prior records for the fused kernel)
f.time_stage('linearize_assemble', 1); f.sync()
for w in range(2):
    Hf, gf = f.read_normal(w, lo, hi - lo); Hu, gu = u.read_normal(w, lo, hi - lo)
    print("window", w, "g max abs diff", np.abs(gf - gu).max(), "at", np.unravel_index(np.abs(gf - gu).argmax(), gf.shape), "scale", np.abs(gu).max())
    for d in range(4):
        D = np.abs(Hf[:, d] - Hu[:, d]); S = np.abs(Hu[:, d]).max()
        k, a, c = np.unravel_index(D.argmax(), D.shape)
        print(f"  block d={d}: max abs diff {D.max():.3e} (scale {S:.3e}) at row {k} entry ({a},{c}); fused {Hf[k, d, a, c]:.6e} unfused {Hu[k, d, a, c]:.6e}")
        bad = np.argwhere(D > 1e-9 * max(S, 1e-300))
        if len(bad):
            rows = sorted(set(bad[:, 0].tolist()))
            print("    bad rows", rows[:12], "..." if len(rows) > 12 else "", "bad (a,c) pattern of first bad row:", sorted(set(map(tuple, bad[bad[:, 0] == rows[0]][:, 1:].tolist())))[:40])
