"""-m gpu: the library's DEFAULTS on windows past the reach of float64 normal equations (vf_engine_opts.refine_min_keyframes =
1536), at lengths and on sequences other than the one BASELINE configs[4] fixture: from IMU dead reckoning, LM in
solves of five trials (what a GraphManager's vf_solve runs) and the reference's Gauss-Newton updates, against the oracle's
refined optimum of the same factors.  What this pins beside convergence itself: the damping persists from one solve to the next
under the non-monotone rule (a solve of five trials that restarted from lambda0 each time never got a 6 000-keyframe window
out of its first excursion, and crept at 8 000), and the windows just above the threshold behave like the ones far above it."""
import numpy as np
import pytest

from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed", [(1600, 11), (2500, 12), (6000, 12)])
def test_defaults_converge_long_windows_from_dead_reckoning(oracle, n, seed):
    seq = synth.make_sequence(seed=seed, n_kf=n)
    prob = helpers.build_problem(oracle, seq)
    ref = helpers.oracle_window(oracle, prob)
    for _ in range(7):
        oracle.gn_step(ref, refine=12)
    start = helpers.ate(prob["states"], ref.states)[0]
    eng = Engine(EngineOpts(windows=1, capacity=n))
    helpers.load_engine(eng, 0, prob)
    assert eng.refine_count() == 12
    hist = []
    for _ in range(3):
        eng.iterate(5)
        hist.append(helpers.ate(eng.get_states(0, 0, n), ref.states)[0])
    lm, ex = eng.read_lm(0), eng.read_excursions(0)
    eng.close()
    eng = Engine(EngineOpts(windows=1, capacity=n))
    helpers.load_engine(eng, 0, prob)
    gn = []
    for _ in range(5):
        eng.isam_step(0.0)
        gn.append(helpers.ate(eng.get_estimate(0, 0, n), ref.states)[0])
    eng.close()
    print(f"n = {n}, seed {seed}: dead reckoning {start:.2f} m from the oracle's refined optimum; LM after 5 / 10 / 15 trials: "
          f"{' '.join(f'{a:.1e}' for a in hist)} m ({lm['accepted']} accepted, {ex[0]} provisional, {lm['rejected']} rejected); Gauss-Newton per update: "
          f"{' '.join(f'{a:.1e}' for a in gn)} m")
    assert start > 0.5 and hist[-1] <= 1e-6 and gn[-1] <= 1e-6 and lm["solve_failures"] == 0 and ex[1] == 0


def test_fixed_lag_updates_of_a_window_that_refines(oracle):
    """A fixed-lag window longer than the threshold: refined solves, the non-monotone rule with its persisting damping, the
    marginal prior's information in the operator (k_jtu applies L p: the prior is kept in information form), the gauge floor
    and the warm start all at once -- 1 700 keyframes, eight marginalised updates of four trials with the appended factor
    preintegrated at the current bias, against the oracle doing the same."""
    from tests.test_gpu_ingest import _engine, _feed
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    n, U, K = 1700, 8, 4
    seq = synth.make_sequence(seed=13, n_kf=n + U + 2)
    eng = _engine(None, [seq], n, U)
    assert eng.refine_count() == 12
    eng.iterate(20)
    prob = helpers.build_problem(oracle, seq)
    ref = helpers.FixedLagOracle(oracle, prob, n, K, init_iterations=20, ingest=(seq, oracle.carla_imu_params()), refine=12, excursion=3)
    a0 = helpers.ate(eng.get_states(0, 0, n), ref.window_states)[0]
    worst = 0.0
    for u in range(1, U + 1):
        eng.ingest_tail(*_feed([seq], n + u - 1))
        eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
        eng.iterate(K)
        ref.update()
        a, r = helpers.ate(eng.get_states(0, u, n), ref.window_states)
        worst = max(worst, a)
        assert r <= 1e-6
    eng.ingest_status()
    lm = eng.read_lm(0)
    got = eng.read_marginal(0)
    dL = np.abs(got["L"] - np.array(ref.marg.L[:]).reshape(27, 27)).max() / np.abs(got["L"]).max()
    print(f"fixed lag, {n} keyframes (refined): batch optimum {a0:.3e} m from the oracle's, worst over {U} updates {worst:.3e} m; cost {lm['cost']:.9f} "
          f"vs {ref.costs[-1]:.9f}; marginal information rel. diff {dL:.1e}; lm {lm}")
    assert a0 <= 1e-6 and worst <= 1e-6 and dL <= 1e-6 and lm["solve_failures"] == 0
    assert abs(lm["cost"] - ref.costs[-1]) <= 1e-8 * ref.costs[-1]
    eng.close()


def test_a_long_window_with_loop_closures_converges_by_the_defaults(oracle):
    """The reference's own use of loop closures: an unbounded graph (lag = 0) that has grown past 1 536 keyframes and takes a
    BetweenFactor between keys far apart (GraphManager.cpp:83-88).  Such a window is refined by default: its far factors are
    rows of the refinement's operator (k_far_apply) and the Woodbury solve is the preconditioner.  No oracle can hold a band
    1 500 keyframes wide, so the check is against the library's other route to the same optimum: the unrefined engine
    (normal equations + Woodbury correction), which at 1 700 keyframes still converges, only slower (contraction 0.16 per
    Gauss-Newton update against 1e-2).  Same start (the refined optimum WITHOUT the closures, so both only have the closures'
    correction to find), 8 Gauss-Newton updates each."""
    from tests.test_gpu_far_factors import _far_record
    n = 1700
    seq = synth.make_sequence(seed=14, n_kf=n)
    prob = helpers.build_problem(oracle, seq)
    ref = helpers.oracle_window(oracle, prob)
    for _ in range(6):
        oracle.gn_step(ref, refine=12)
    prob = dict(prob, states=ref.states.copy())
    rng = np.random.default_rng(15)
    closures = ((100, 1650), (400, 1200))
    fa, fb = np.array([c[0] for c in closures], dtype=np.int32), np.array([c[1] for c in closures], dtype=np.int32)
    far = np.stack([_far_record(seq, a, b, rng, cov=1e-4, noise=(1e-4, 1e-3)) for a, b in closures])
    out = {}
    for name, opts in (("defaults", {}), ("unrefined", dict(refine_iterations=0, lm_excursion=0))):
        eng = Engine(EngineOpts(windows=1, capacity=n, **opts))
        helpers.load_engine(eng, 0, prob)
        eng.set_extra_between(0, fa, fb, far)
        assert eng.refine_count() == (12 if name == "defaults" else 0)
        hist = []
        for _ in range(8):
            eng.isam_step(0.0)
            hist.append(eng.get_estimate(0, 0, n))
        out[name] = hist
        eng.close()
    moved = helpers.ate(out["defaults"][-1], prob["states"])[0]
    d = [helpers.ate(a, out["defaults"][-1])[0] for a in out["defaults"][:-1]]
    du = [helpers.ate(a, out["defaults"][-1])[0] for a in out["unrefined"]]
    print(f"{n} keyframes, closures {closures}: they move the optimum by {moved:.3e} m; distance to the refined engine's 8th update -- "
          f"refined, updates 1..7: {' '.join(f'{x:.1e}' for x in d)}; unrefined (Woodbury), updates 1..8: {' '.join(f'{x:.1e}' for x in du)}")
    assert moved > 1e-3 and d[3] <= 1e-7 and du[-1] <= 1e-6
