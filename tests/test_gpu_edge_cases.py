"""-m gpu: edge cases of the hot path (tiny / ragged windows, missing factors, sliding, failure
flags, determinism), each against the oracle on the same inputs."""
import numpy as np
import pytest

from tests import helpers
from vil_sensor_fusion_amd import synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS

pytestmark = pytest.mark.gpu


def _solve_both(oracle, prob, lo, hi, iters=4, eng=None, window=0):
    from vil_sensor_fusion_amd import Engine, EngineOpts
    own = eng is None
    if own:
        eng = Engine(EngineOpts(windows=1, capacity=max(64, prob["n"])))
    p = dict(prob)
    p["prior"] = synth.prior_record(prob["states"][lo], REFERENCE_PRIOR_SIGMAS)
    helpers.load_engine(eng, window, p, lo=lo, hi=hi)
    eng.iterate(iters)
    xs = eng.get_states(window, lo, hi - lo)
    win = helpers.oracle_window(oracle, p, lo=lo, hi=hi)
    costs, acc, _ = win.lm(iterations=iters)
    return eng, xs, win, costs


@pytest.mark.parametrize("n", [2, 3, 4, 5, 7, 9])
def test_tiny_windows(oracle, n):
    seq = synth.make_sequence(21, 16)
    prob = helpers.build_problem(oracle, seq, perturb=0.01)
    eng, xs, win, costs = _solve_both(oracle, prob, 0, n)
    ate, rot = helpers.ate(xs, win.states)
    assert ate <= 1e-6 and rot <= 1e-6, (n, ate, rot)
    assert abs(eng.read_lm(0)["cost"] - costs[-1]) <= 1e-6 * max(costs[-1], 1e-9)


def test_window_in_the_middle_of_the_slots(oracle):
    """lo > 0: between factors reaching back before lo must be ignored (a < lo)."""
    seq = synth.make_sequence(22, 120)
    prob = helpers.build_problem(oracle, seq, perturb=0.01)
    eng, xs, win, costs = _solve_both(oracle, prob, 37, 101)
    ate, rot = helpers.ate(xs, win.states)
    print("middle window ATE", ate)
    assert ate <= 1e-6 and rot <= 1e-6


@pytest.mark.parametrize("vio,lidar", [(False, True), (True, False)])
def test_single_odometry_source(oracle, vio, lidar):
    """BASELINE configs[0]: IMU + LOAM between factors only (and the converse)."""
    seq = synth.make_sequence(23, 90, vio=vio, lidar=lidar)
    prob = helpers.build_problem(oracle, seq, perturb=0.01)
    assert prob["btw_a"].size > 60
    eng, xs, win, costs = _solve_both(oracle, prob, 0, 90, iters=5)
    ate, rot = helpers.ate(xs, win.states)
    assert ate <= 1e-6 and rot <= 1e-6


def test_config1_900_keyframe_clip_lidar_only(oracle):
    """BASELINE configs[0] at the size SURVEY 8(d) names: the 30 s Carla-like clip (900 keyframes), IMU + LiDAR between
    factors only -- the HIP path against the CPU oracle (which is this config's own 'CPU fixed-lag, plumbing' leg)."""
    n = 900
    seq = synth.make_sequence(25, n, vio=False, lidar=True)
    prob = helpers.build_problem(oracle, seq, perturb=0.01)
    assert prob["btw_a"].size > 250
    eng, xs, win, costs = _solve_both(oracle, prob, 0, n, iters=10)      # (from 1 cm / 0.01 rad off: both must have converged)
    ate, rot = helpers.ate(xs, win.states)
    gt_ate = helpers.ate(xs, seq.gt_states)[0]
    print(f"C1, 900 keyframes, IMU + LiDAR only: ATE vs oracle {ate:.3e} m, rot {rot:.3e} rad; vs ground truth {gt_ate:.3e} m; "
          f"oracle cost {costs[0]:.3e} -> {costs[-1]:.3e}")
    assert ate <= 1e-6 and rot <= 1e-6 and costs[-1] < costs[0]


def test_imu_only_and_dropped_between_factors(oracle):
    """No between factor at all (IMU chain + priors), then every third factor dropped
    (what the degeneracy filter does to LOAM odometry)."""
    seq = synth.make_sequence(24, 60)
    prob = helpers.build_problem(oracle, seq, perturb=0.005)
    for keep in (np.zeros(prob["btw_a"].size, bool), np.arange(prob["btw_a"].size) % 3 != 0):
        p = dict(prob)
        p["btw_a"], p["btw_b"], p["btw"] = prob["btw_a"][keep], prob["btw_b"][keep], prob["btw"][keep]
        from vil_sensor_fusion_amd import Engine, EngineOpts
        eng = Engine(EngineOpts(windows=1, capacity=64))
        helpers.load_engine(eng, 0, p)
        eng.iterate(4)
        win = helpers.oracle_window(oracle, p)
        win.lm(iterations=4)
        ate, rot = helpers.ate(eng.get_states(0, 0, 60), win.states)
        assert ate <= 1e-6 and rot <= 1e-6


def test_ragged_batch_and_determinism(oracle):
    """Windows of different lengths / offsets in one batch; two runs are bitwise identical."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    ranges = [(0, 50), (5, 64), (10, 13), (0, 2), (20, 61)]
    seqs = [synth.make_sequence(30 + i, 64) for i in range(len(ranges))]
    probs = [helpers.build_problem(oracle, s, perturb=0.01) for s in seqs]
    outs = []
    for rep in range(2):
        eng = Engine(EngineOpts(windows=len(ranges), capacity=64))
        for w, ((lo, hi), p) in enumerate(zip(ranges, probs)):
            q = dict(p); q["prior"] = synth.prior_record(p["states"][lo], REFERENCE_PRIOR_SIGMAS)
            helpers.load_engine(eng, w, q, lo=lo, hi=hi)
        eng.iterate(4)
        outs.append([eng.get_states(w, lo, hi - lo) for w, (lo, hi) in enumerate(ranges)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    for w, ((lo, hi), p) in enumerate(zip(ranges, probs)):
        q = dict(p); q["prior"] = synth.prior_record(p["states"][lo], REFERENCE_PRIOR_SIGMAS)
        win = helpers.oracle_window(oracle, q, lo=lo, hi=hi)
        win.lm(iterations=4)
        ate, rot = helpers.ate(outs[0][w], win.states)
        assert ate <= 1e-6 and rot <= 1e-6, (w, ate, rot)


def test_slide_reanchor_matches_oracle(oracle):
    """Fallback fixed-lag update: slide (predict new keyframe, drop oldest, re-anchor the prior at
    the current estimate) + K LM trials, three times, against the oracle doing the same."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    N, S = 40, 3
    seq = synth.make_sequence(41, N + S)
    prob = helpers.build_problem(oracle, seq)
    eng = Engine(EngineOpts(windows=1, capacity=64))
    helpers.load_engine(eng, 0, prob, lo=0, hi=N)
    eng.iterate(4)
    states = prob["states"].copy()
    win = helpers.oracle_window(oracle, prob, lo=0, hi=N)
    win.lm(iterations=4)
    states[:N] = win.states
    for s in range(1, S + 1):
        eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=False)
        eng.iterate(4)
        # oracle: predict the new keyframe from the current estimate, re-anchor, solve
        states[N + s - 1] = oracle.predict(prob["imu"][N + s - 1], prob["gravity"], states[N + s - 2])
        p = dict(prob); p["states"] = states
        p["prior"] = synth.prior_record(states[s], REFERENCE_PRIOR_SIGMAS)
        win = helpers.oracle_window(oracle, p, lo=s, hi=N + s)
        win.lm(iterations=4)
        states[s:N + s] = win.states
        ate, rot = helpers.ate(eng.get_states(0, s, N), win.states)
        print(f"slide {s}: ATE {ate:.3e} rot {rot:.3e}")
        assert ate <= 1e-6 and rot <= 1e-6


def test_indeterminate_system_is_flagged(oracle):
    """A prior with absurdly large sigma on an IMU-only chain of 2 keyframes is still PD; a
    non-positive pivot is produced by feeding a negative-definite 'information': emulate by
    corrupting H through a NaN state -> every trial must be rejected and counted."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    seq = synth.make_sequence(50, 16)
    prob = helpers.build_problem(oracle, seq)
    eng = Engine(EngineOpts(windows=1, capacity=64))
    bad = dict(prob); bad["states"] = prob["states"].copy(); bad["states"][5, 4] = np.nan
    helpers.load_engine(eng, 0, bad)
    eng.iterate(3)
    lm = eng.read_lm(0)
    assert lm["accepted"] == 0 and lm["rejected"] == 3     # NaN cost never accepted (NaN compares false)
    assert lm["solve_failures"] >= 1


def _oracle_window(oracle, prob, states, lo, hi, marg, with_prior):
    m = (prob["btw_a"] >= lo) & (prob["btw_b"] < hi)
    ks = np.arange(lo + 1, hi)
    pk = np.array([0], dtype=np.int32) if with_prior else np.zeros(0, dtype=np.int32)
    pd = prob["prior"].reshape(1, -1) if with_prior else np.zeros((0, 31))
    w = oracle.Window(states[lo:hi], ks - 1 - lo, ks - lo, prob["imu"][lo + 1:hi], prob["btw_a"][m] - lo,
                      prob["btw_b"][m] - lo, prob["btw"][m], pk, pd, prob["gravity"])
    if marg is not None:
        w.set_marg(marg)
    return w


@pytest.mark.parametrize("N,S,chunks", [(40, 5, 0), (160, 25, 0), (160, 25, 1)])
def test_marginal_prior_and_fixed_lag_slides_match_oracle(oracle, N, S, chunks):
    """bench.py's update path: marginalise the oldest keyframe (Schur complement into a dense
    prior), slide, K LM trials -- S times -- against the oracle doing the same; the marginal
    prior itself (27x27 information, gradient, linearisation states) is compared after every slide.
    chunks = 0: the partitioned solve (the marginal prior sits in the first chunk), 1: whole-window sweeps."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    seq = synth.make_sequence(43, N + S)
    prob = helpers.build_problem(oracle, seq)
    eng = Engine(EngineOpts(windows=2, capacity=N + S + 4, chunks=chunks))     # second window: ragged companion
    helpers.load_engine(eng, 0, prob, lo=0, hi=N)
    helpers.load_engine(eng, 1, prob, lo=0, hi=N - 7)
    eng.iterate(4)
    states = prob["states"].copy()
    win = _oracle_window(oracle, prob, states, 0, N, None, True)
    win.lm(iterations=4)
    states[:N] = win.states
    marg = None
    for s in range(1, S + 1):
        eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
        got = eng.read_marginal(0)
        prev = _oracle_window(oracle, prob, states, s - 1, N + s - 1, marg, s == 1)
        marg = prev.marginalize(0)
        exp = marg.arrays()
        assert got["on"] == 1
        scale = np.abs(exp["L"]).max()
        np.testing.assert_allclose(got["L"], exp["L"], atol=1e-9 * scale)
        if s <= 5:
            # eta = information x (state differences): once the two LM trajectories have drifted apart by 1e-10 m
            # (later slides of the long runs) it can only be compared through the trajectories themselves
            np.testing.assert_allclose(got["eta"], exp["eta"], atol=1e-9 * max(np.abs(exp["eta"]).max(), 1e-6 * scale))
        np.testing.assert_allclose(got["xbar"], exp["xbar"], atol=1e-12 if s <= 5 else 1e-8)
        eng.iterate(4)
        marg.k0 = 0
        states[N + s - 1] = oracle.predict(prob["imu"][N + s - 1], prob["gravity"], states[N + s - 2])
        win = _oracle_window(oracle, prob, states, s, N + s, marg, False)
        costs, acc, _ = win.lm(iterations=4)
        states[s:N + s] = win.states
        ate, rot = helpers.ate(eng.get_states(0, s, N), win.states)
        lm = eng.read_lm(0)
        print(f"marginalised slide {s}: ATE {ate:.3e} rot {rot:.3e} cost gpu {lm['cost']:.9e} oracle {costs[-1]:.9e}")
        assert ate <= 1e-6 and rot <= 1e-6
        assert abs(lm["cost"] - costs[-1]) <= 1e-6 * max(abs(costs[-1]), 1e-9)


def test_graph_manager_fixed_lag_marginalises(oracle):
    """GraphManager with lag: the window never exceeds `lag` keyframes and the estimate stays
    close to the unbounded (full-history) smoother on the same stream."""
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    n = 70
    seq = synth.make_sequence(44, n)
    traj_t = synth.IMU_PHASE + np.arange(0, int((seq.kf_time[-1] + 0.5) * synth.IMU_RATE)) / synth.IMU_RATE
    traj = synth.Trajectory(seq.seed, seq.kf_time[-1] + 1.0)
    rng = np.random.default_rng([seq.seed, 0xBEEF])
    acc = traj.specific_force(traj_t) + rng.normal(size=(traj_t.size, 3)) * synth.IMU_NOISE
    gyr = traj.body_rate(traj_t) + rng.normal(size=(traj_t.size, 3)) * synth.IMU_NOISE
    out = {}
    for name, lag in (("full", 0), ("lag", 24)):
        gm = GraphManager(capacity=128, iterations=5, lag=lag)
        i_imu = 0
        for k in range(1, n):
            while i_imu < traj_t.size and traj_t[i_imu] <= seq.kf_time[k] + 0.01:
                gm.addIMUMeasurement(traj_t[i_imu], acc[i_imu], gyr[i_imu]); i_imu += 1
            gm.reserveNode(seq.kf_time[k])
            for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov):
                if b == k and a >= 1:
                    gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
            if k % 2 == 0:
                gm.solve()
        gm.solve()
        out[name] = gm.getState()
    (qf, tf), vf_, bf = out["full"]
    (ql, tl), vl, bl = out["lag"]
    dpos = np.linalg.norm(tf - tl)
    print("fixed-lag (24) vs full-history smoother, last pose difference [m]:", dpos)
    assert dpos < 5e-3


def test_engine_compaction_preserves_the_problem(oracle):
    """vf_engine_compact moves the live keyframes down by whole AoSoA tiles; solving after the
    move gives bitwise the same states as solving without it."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    seq = synth.make_sequence(45, 200)
    prob = helpers.build_problem(oracle, seq, perturb=0.01)
    lo, hi = 70, 190
    outs = []
    for compact in (False, True):
        eng = Engine(EngineOpts(windows=2, capacity=256))
        for w in range(2):
            q = dict(prob); q["prior"] = synth.prior_record(prob["states"][lo], REFERENCE_PRIOR_SIGMAS)
            helpers.load_engine(eng, w, q, lo=lo, hi=hi - 5 * w)
        eng.iterate(2)
        eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)     # a marginal prior must survive the move too
        shift = 0
        if compact:
            shift = 64
            eng.compact(shift)
        eng.iterate(3)
        outs.append(eng.get_states(0, lo + 1 - shift, hi - lo))
    np.testing.assert_array_equal(outs[0], outs[1])


def test_graph_manager_runs_past_its_capacity_with_a_lag():
    """lag mode + compaction: 330 keyframes through a 192-slot GraphManager give the same final
    state as through a 512-slot one (no compaction needed there)."""
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    n = 330
    seq = synth.make_sequence(46, n)
    traj_t = synth.IMU_PHASE + np.arange(0, int((seq.kf_time[-1] + 0.5) * synth.IMU_RATE)) / synth.IMU_RATE
    traj = synth.Trajectory(seq.seed, seq.kf_time[-1] + 1.0)
    rng = np.random.default_rng([seq.seed, 0xBEEF])
    acc = traj.specific_force(traj_t) + rng.normal(size=(traj_t.size, 3)) * synth.IMU_NOISE
    gyr = traj.body_rate(traj_t) + rng.normal(size=(traj_t.size, 3)) * synth.IMU_NOISE
    finals = []
    for cap in (512, 192):
        gm = GraphManager(capacity=cap, iterations=3, lag=40)
        i_imu = 0
        for k in range(1, n):
            while i_imu < traj_t.size and traj_t[i_imu] <= seq.kf_time[k] + 0.01:
                gm.addIMUMeasurement(traj_t[i_imu], acc[i_imu], gyr[i_imu]); i_imu += 1
            assert gm.reserveNode(seq.kf_time[k]) == k
            for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov):
                if b == k and a >= 1:
                    gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
            if k % 3 == 0:
                gm.solve()
        gm.solve()
        (q, t), v, b = gm.getState()
        finals.append(np.concatenate([q, t, v, b]))
    np.testing.assert_array_equal(finals[0], finals[1])


def test_lm_convergence_exit_matches_oracle(oracle):
    """Optional LM termination (vf_engine_set_convergence = GTSAM's LM tolerances, applied to every trial
    whose cost change is inside the tolerance): the converged window stops taking trials; same number of
    trials, same trajectory as the oracle under the same rule.
    A second, harder window in the same engine keeps iterating (the flag is per window)."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    n, iters = 120, 10
    eng = Engine(EngineOpts(windows=2, capacity=n + 8))
    probs = []
    for w, perturb in enumerate((0.001, 0.05)):
        seq = synth.make_sequence(seed=40 + w, n_kf=n)
        prob = helpers.build_problem(oracle, seq, perturb=perturb)
        helpers.load_engine(eng, w, prob)
        probs.append(prob)
    eng.set_convergence(1e-5, 1e-5)
    eng.iterate(iters)
    trials = []
    for w in range(2):
        win = helpers.oracle_window(oracle, probs[w])
        costs, acc, _ = win.lm(iterations=iters, rel_tol=1e-5, abs_tol=1e-5)
        ran = int((acc >= 0).sum())
        lm = eng.read_lm(w)
        a, r = helpers.ate(eng.get_states(w, 0, n), win.states)
        print(f"window {w}: trials gpu {lm['accepted'] + lm['rejected']} oracle {ran}; ATE {a:.3e}")
        assert lm["accepted"] + lm["rejected"] == ran
        # (whether the terminating trial itself is accepted is decided by the last bit of the cost: not compared)
        assert abs(lm["accepted"] - int((acc == 1).sum())) <= 1
        assert a <= 1e-6 and r <= 1e-6
        trials.append(ran)
    assert trials[0] < iters                 # the easy window stopped early
    # switched off again: exactly `iters` trials
    eng.set_convergence(0.0, 0.0)
    before = [eng.read_lm(w) for w in range(2)]
    eng.iterate(3)
    for w in range(2):
        lm = eng.read_lm(w)
        assert lm["accepted"] + lm["rejected"] - before[w]["accepted"] - before[w]["rejected"] == 3
    eng.close()


def test_read_panels_layout_known_answer(oracle):
    """vf_engine_read_panels returns [43][16] per keyframe whatever the packed device layout: for the FIRST keyframe of a
    window the pivot block is H00 + lambda I itself, so rows 28..42 (L^-T, upper triangular) satisfy
    U U^T (H00 + lambda I) = I, and row 27 is y = L^-1 (-g0)."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    n = 40
    seq = synth.make_sequence(seed=3, n_kf=n)
    prob = helpers.build_problem(oracle, seq, perturb=0.01)
    eng = Engine(EngineOpts(windows=1, capacity=n, chunks=1))
    helpers.load_engine(eng, 0, prob)
    eng.linearize(0)
    eng.decide(init=True)
    eng.assemble()
    eng.solve()
    lam = eng.read_lm(0)["lam"]
    H, g = eng.read_normal(0, 0, 1)
    P = eng.read_panels(0, 0, 1)[0]
    U = P[28:43, :15]
    assert np.all(np.tril(U, -1) == 0.0)
    A = H[0, 0] + lam * np.eye(15)
    A = np.tril(A) + np.tril(A, -1).T
    np.testing.assert_allclose(U @ U.T @ A, np.eye(15), atol=1e-9)
    np.testing.assert_allclose(U.T @ (-g[0]), P[27, :15], rtol=1e-9, atol=1e-12)
    assert np.all(P[:, 15] == 0.0)


@pytest.mark.parametrize("chunks", [1, 0])
def test_failed_window_does_not_disturb_its_neighbours(oracle, chunks):
    """A non-positive pivot is no longer replaced inside the pivot chain: the window's factorisation runs on with NaN
    and is flagged once per step.  The NaN must stay inside that window: its neighbours in the batch give bit for bit
    what they give in a clean batch, and the failed window keeps its states (every trial rejected)."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    n = 48
    probs = [helpers.build_problem(oracle, synth.make_sequence(60 + w, n), perturb=0.01) for w in range(3)]
    clean = Engine(EngineOpts(windows=3, capacity=n, chunks=chunks))
    dirty = Engine(EngineOpts(windows=3, capacity=n, chunks=chunks))
    for w in range(3):
        helpers.load_engine(clean, w, probs[w])
        p = probs[w]
        if w == 1:
            p = dict(p); p["states"] = p["states"].copy(); p["states"][7, 5] = np.nan
        helpers.load_engine(dirty, w, p)
    before = dirty.get_states(1, 0, n)
    clean.iterate(4)
    dirty.iterate(4)
    for w in (0, 2):
        assert np.array_equal(clean.get_states(w, 0, n), dirty.get_states(w, 0, n))
        assert clean.read_lm(w) == dirty.read_lm(w)
    lm = dirty.read_lm(1)
    assert lm["accepted"] == 0 and lm["rejected"] == 4 and lm["solve_failures"] >= 1
    after = dirty.get_states(1, 0, n)
    assert np.array_equal(np.isnan(before), np.isnan(after)) and np.array_equal(before[~np.isnan(before)], after[~np.isnan(after)])


def test_results_do_not_depend_on_the_batch(oracle):
    """The same window gives bit for bit the same states whether it is alone in an engine or one of several copies
    in a batch (one sweep per window: chunks=1), and all copies agree with each other."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    n = 96
    prob = helpers.build_problem(oracle, synth.make_sequence(71, n), perturb=0.01)
    alone = Engine(EngineOpts(windows=1, capacity=n, chunks=1))
    helpers.load_engine(alone, 0, prob)
    alone.iterate(4)
    ref = alone.get_states(0, 0, n)
    many = Engine(EngineOpts(windows=9, capacity=n, chunks=1))
    for w in range(9):
        helpers.load_engine(many, w, prob)
    many.iterate(4)
    for w in range(9):
        assert np.array_equal(many.get_states(w, 0, n), ref), w
        assert many.read_lm(w) == alone.read_lm(0)


def test_engine_grow_carries_the_problem_over(oracle):
    """vf_engine_grow on a batch engine with ragged windows and a marginal prior: after growing, a solve gives bitwise what
    an engine created at the larger capacity gives (states, factor records, priors, marginal priors, ranges and LM
    counters are carried over on the device; linearisations are recomputed)."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    from vil_sensor_fusion_amd._lib import VilFusionError
    n = 100
    seq = synth.make_sequence(52, n + 8)
    prob = helpers.build_problem(oracle, seq, perturb=0.01)
    ranges = [(0, n), (0, 64), (5, 90)]

    def prepared(capacity):
        eng = Engine(EngineOpts(windows=3, capacity=capacity, chunks=1))
        for w, (lo, hi) in enumerate(ranges):
            helpers.load_engine(eng, w, prob, lo=lo, hi=hi)
        eng.iterate(3)
        eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)       # leaves a marginal prior and a predicted keyframe
        eng.iterate(2)
        return eng

    small, big = prepared(128), prepared(320)
    small.grow(320)
    assert small.capacity == 320
    for e in (small, big):
        e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
        e.iterate(3)
    for w, (lo, hi) in enumerate(ranges):
        a, b = small.get_states(w, lo + 2, hi - lo), big.get_states(w, lo + 2, hi - lo)
        np.testing.assert_array_equal(a, b)
        assert small.read_lm(w) == big.read_lm(w)
        ma, mb = small.read_marginal(w), big.read_marginal(w)
        assert ma["on"] == mb["on"] == 1
        np.testing.assert_array_equal(ma["L"], mb["L"])
        np.testing.assert_array_equal(small.get_imu(w, hi, 2), big.get_imu(w, hi, 2))
    with pytest.raises(VilFusionError):
        small.grow(320)                     # must exceed the current capacity
    small.close(); big.close()


def test_gauge_floor_on_the_device_matches_the_oracle_and_keeps_long_runs_solvable(oracle):
    """vf_engine_opts.gauge_floor in k_marginalize against vfo_marginalize_floor (tests/test_oracle_marginalization.py says what
    it is for): 900 fixed-lag updates of 64-keyframe windows, with compaction cycles.  With the floor (the default) device and
    oracle carry the same marginal prior (information to 1e-6 of its largest entry), its gauge information sits at
    floor * n / 3 or above, nothing fails and next to nothing is rejected; with the floor off the device's own window ends
    with its global translation information below that and falling (the 1 000-keyframe windows of tools/soak.py reach the
    rounding level after ~2 000 updates and fail after ~3 500), which is what the floor is there to prevent."""
    from tests.test_gpu_ingest import _feed
    from vil_sensor_fusion_amd import Engine, EngineOpts
    n, U, K = 64, 900, 4
    seqs = [synth.make_sequence(seed=430 + i, n_kf=n + U + 2) for i in range(2)]
    prm = oracle.carla_imu_params()
    res = {}
    for floor in (None, 0.0):
        cap = n + 128
        eng = Engine(EngineOpts(windows=len(seqs), capacity=cap, gauge_floor=floor))
        for w, seq in enumerate(seqs):
            eng.preintegrate(w, 1, seq.imu_off[1:n + 1], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
            m = seq.btw_b < n
            eng.set_between(w, seq.btw_a[m], seq.btw_b[m], synth.between_records(seq)[m])
            eng.set_states(w, 0, seq.gt_states[0].reshape(1, 16))
            eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
            eng.set_range(w, 0, 1)
        eng.predict(-1, 1, n - 1)
        for w in range(len(seqs)):
            eng.set_range(w, 0, n)
        eng.iterate(60)
        refs = None
        if floor is None:
            refs = [helpers.FixedLagOracle(oracle, helpers.build_problem(oracle, s), n, K, init_iterations=60, ingest=(s, prm)) for s in seqs]
        base = 0
        for u in range(1, U + 1):
            k = n + u - 1
            if k - base >= cap:
                shift = (k - n - base) // 64 * 64
                eng.compact(shift)
                base += shift
            off, steps, cov, a, rec = _feed(seqs, k)
            eng.ingest_tail(off, steps, cov, np.where(a >= 0, a - base, -1).astype(np.int32), rec)
            eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
            eng.iterate(K)
            if refs:
                for r in refs:
                    r.update()
        eng.ingest_status()
        lms = [eng.read_lm(w) for w in range(len(seqs))]
        mp = [eng.read_marginal(w) for w in range(len(seqs))]
        st = [eng.get_states(w, U - base, n) for w in range(len(seqs))]
        res[floor] = dict(lm=lms, mp=mp, states=st, refs=refs)
        eng.close()

    def gauge_info(mp):
        """eigenvalues of G^T L G, G = the prior's orthonormalised translation / yaw directions (as k_marginalize builds them)"""
        x, L = mp["xbar"], mp["L"]
        G = np.zeros((27, 4))
        for j in range(3):
            R = synth.quat_to_rot(x[j, :4])
            o = 0 if j == 0 else 15 + 6 * (j - 1)
            G[o + 3:o + 6, :3] = R.T
            G[o:o + 3, 3] = R.T @ np.array([0.0, 0.0, 1.0])
            G[o + 3:o + 6, 3] = R.T @ np.cross([0.0, 0.0, 1.0], x[j, 4:7] - x[0, 4:7])
        G[6:9, 3] = np.cross([0.0, 0.0, 1.0], x[0, 7:10])
        Q, _ = np.linalg.qr(G)
        return np.linalg.eigvalsh(Q.T @ (0.5 * (L + L.T)) @ Q)

    on, off_ = res[None], res[0.0]
    want = oracle.prior_gauge_floor(n)
    for w in range(len(seqs)):
        ref = on["refs"][w]
        Lo = np.array(ref.marg.L[:]).reshape(27, 27)
        dL = np.abs(on["mp"][w]["L"] - Lo).max() / np.abs(Lo).max()
        a, r = helpers.ate(on["states"][w], ref.window_states)
        gi_on, gi_off = gauge_info(on["mp"][w]), gauge_info(off_["mp"][w])
        print(f"window {w}: {U} updates; with the floor: marginal information vs oracle {dL:.1e}, ATE {a:.2e} m, gauge information {gi_on} (floor {want:.2e}), "
              f"lm {on['lm'][w]}; without: gauge information {gi_off}, lm {off_['lm'][w]}")
        # (unaligned ATE between two float64 runs of a gauge-free smoother is a random walk of the gauge: 1e-5 m by now)
        assert dL <= 1e-6 and a <= 1e-4 and r <= 1e-5
        assert gi_on.min() >= 0.98 * want and on["lm"][w]["solve_failures"] == 0 and on["lm"][w]["rejected"] <= 20
        assert abs(on["lm"][w]["cost"] - ref.costs[-1]) <= 1e-6 * ref.costs[-1]
        assert gi_off.min() < 0.6 * want                        # without the floor the information has decayed below it (and goes on decaying)
