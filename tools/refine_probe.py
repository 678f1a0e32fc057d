"""GPU probe: BASELINE configs[4] (10 000-pose window, seed 4242) from IMU dead reckoning through reference-compat updates
(undamped Gauss-Newton, vf_engine_isam_step) with and without the refined solve, against tests/golden/qr_twin_10k.npz."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402
from vil_sensor_fusion_amd import Engine, EngineOpts, synth  # noqa: E402
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS  # noqa: E402

F = np.load(os.path.join(ROOT, "tests", "golden", "qr_twin_10k.npz"))
n = int(F["n"])
seq = synth.make_sequence(seed=int(F["seed"]), n_kf=n)


def load(**opts):
    eng = Engine(EngineOpts(windows=1, capacity=n, **opts))
    eng.preintegrate(0, 1, seq.imu_off[1:n + 1], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
    eng.set_between(0, seq.btw_a, seq.btw_b, synth.between_records(seq))
    eng.set_states(0, 0, seq.gt_states[:1])
    eng.set_prior(0, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
    eng.set_range(0, 0, 1)
    eng.predict(0, 1, n - 1)
    eng.set_range(0, 0, n)
    return eng


def gn(tag, steps=6, **opts):
    eng = load(**opts)
    print(f"== {tag}: form {eng.solve_form()}, refine count {eng.refine_count()}", flush=True)
    for it in range(steps):
        t0 = time.perf_counter()
        eng.isam_step(0.0)
        est = eng.get_estimate(0, 0, n)
        dt = time.perf_counter() - t0
        a, r = helpers.ate(est, F["states"])
        print(f"  GN step {it}: ATE vs QR optimum {a:.3e} m, rot {r:.3e} rad, refine (corrections, reduction) {eng.read_refine(0)}, {dt * 1e3:.1f} ms", flush=True)
    eng.close()


def lm(tag, trials, **opts):
    eng = load(**opts)
    print(f"== {tag}: form {eng.solve_form()}, refine count {eng.refine_count()}", flush=True)
    eng.reset_lambda()
    eng.linearize(0)
    eng.decide(True)
    for it in range(trials):
        eng.assemble(), eng.solve(), eng.retract(), eng.linearize(1), eng.decide(False)
        a, _ = helpers.ate(eng.get_states(0, 0, n), F["states"])
        print(f"  LM trial {it}: {eng.read_lm(0)} excursions {eng.read_excursions(0)} ATE {a:.3e}", flush=True)
    eng.close_excursions()
    a, _ = helpers.ate(eng.get_states(0, 0, n), F["states"])
    print(f"  closed: {eng.read_lm(0)} ATE {a:.3e}")
    eng.close()


def trace(steps=3, corrections=14):
    """the refinement correction by correction (staged calls): how fast res . M^-1 res falls, and what each correction is worth in
    the Gauss-Newton update it belongs to (ATE of the estimate one would get by stopping there)"""
    eng = load(refine_iterations=corrections)
    for it in range(steps):
        eng.gn_begin(0.0)
        eng.assemble(), eng.solve_local(), eng.solve_global()
        theta = eng.get_states(0, 0, n)
        eng.refine_begin()
        row = []
        for c in range(corrections):
            eng.solve_local(), eng.solve_global(), eng.refine_step()
            k, red = eng.read_refine(0)
            row.append(f"{c + 1}:{red:.1e}" + ("" if k == c + 1 else "(stopped)"))
        eng.refine_end()
        eng.retract()
        a, _ = helpers.ate(eng.get_estimate(0, 0, n), F["states"])
        print(f"  GN update {it}: reduction of res.M^-1 res after each correction: {' '.join(row)}; ATE after the update {a:.3e} m", flush=True)
    eng.close()


if __name__ == "__main__":
    what = sys.argv[1:] or ["gn"]
    if "gn" in what:
        gn("partitioned, refined (auto)")
        gn("one sweep per window, refined", chunks=1, sweep_two_sided_max=0)
        gn("partitioned, NOT refined", steps=4, refine_iterations=0)
    if "gnp" in what:
        gn("partitioned, refined (auto)", steps=4)
    if "trace" in what:
        trace()
    if "lm" in what:
        lm("LM, partitioned, refined, excursions (the defaults for a window this long)", 20)
        lm("LM, partitioned, refined, classical accept rule", 20, lm_excursion=0)
