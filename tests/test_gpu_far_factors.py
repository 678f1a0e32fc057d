"""-m gpu: between factors the band cannot hold.  iSAM2 takes a BetweenFactor<Pose3> on any pair of keys
(GraphManager.cpp:83-88); the device's banded solver holds spans <= 3 and one factor per end key.  Anything else -- a span-6
factor, a loop closure between keyframes far apart, a second factor ending at a key -- is a "far" factor
(vf_engine_set_extra_between, routed there by vf_add_between): its J^T r goes into g, its J^T J is applied as a low-rank
(Woodbury) correction around the band solver.  Checked against the CPU oracle, which assembles a band as wide as the widest
factor and so treats these factors like any other."""
import numpy as np
import pytest

from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, VilFusionError, synth

pytestmark = pytest.mark.gpu


def _far_record(seq, a, b, rng, cov=0.05, noise=(5e-4, 5e-3)):
    """relative pose of keyframes a -> b from the ground truth + noise, as a 28-double record (isotropic covariance)"""
    Ra, Rb = synth.quat_to_rot(seq.gt_states[a, :4]), synth.quat_to_rot(seq.gt_states[b, :4])
    Rab = Ra.T @ Rb @ synth.so3_exp(rng.normal(size=3) * noise[0])
    tab = Ra.T @ (seq.gt_states[b, 4:7] - seq.gt_states[a, 4:7]) + rng.normal(size=3) * noise[1]
    rec = np.zeros(28)
    rec[0:4], rec[4:7] = synth.rot_to_quat(Rab), tab
    iu = np.triu_indices(6)
    rec[7 + np.nonzero(iu[0] == iu[1])[0]] = 1.0 / np.sqrt(cov)
    return rec


FAR = [(40, 46), (20, 70), (21, 71)]         # a span-6 factor whose end key 46 also has its band factor; a loop-closure pair
#                                              (span 50: the oracle's dense band costs span^2 -- at span 110 it was half of the GPU suite's time)


@pytest.mark.parametrize("form", ["partitioned", "one_wave_sweep", "refined"])
def test_far_factors_match_the_oracle(oracle, form):
    n = 200
    seq = synth.make_sequence(seed=91, n_kf=n)
    prob = helpers.build_problem(oracle, seq, perturb=0.003)
    rng = np.random.default_rng(5)
    far_rec = np.array([_far_record(seq, a, b, rng) for a, b in FAR])
    fa, fb = np.array([a for a, _ in FAR], dtype=np.int32), np.array([b for _, b in FAR], dtype=np.int32)
    opts = dict(chunks=1, sweep_two_sided_max=0) if form == "one_wave_sweep" else {}
    if form == "refined":
        # the refined solve (the default of windows longer than 1 536 keyframes, forced here): the far factors are rows of the
        # operator of its conjugate gradients too, the Woodbury solve is their preconditioner
        opts = dict(refine_iterations=8, lm_excursion=0)
    # three windows on one engine: all three far factors, none, the loop-closure pair only
    eng = Engine(EngineOpts(windows=3, capacity=n, **opts))
    for w in range(3):
        helpers.load_engine(eng, w, prob)
    eng.set_extra_between(0, fa, fb, far_rec)
    eng.set_extra_between(2, fa[1:], fb[1:], far_rec[1:])
    eng.iterate(20)
    outs = []
    for w, sel in ((0, [0, 1, 2]), (1, []), (2, [1, 2])):
        p = dict(prob, btw_a=np.concatenate([prob["btw_a"], fa[sel]]).astype(np.int32),
                 btw_b=np.concatenate([prob["btw_b"], fb[sel]]).astype(np.int32), btw=np.vstack([prob["btw"], far_rec[sel]]))
        win = helpers.oracle_window(oracle, p)
        assert win.bandwidth() == (50 if sel else 3)
        costs, _, _ = win.lm(iterations=20)
        got = eng.get_states(w, 0, n)
        a, r = helpers.ate(got, win.states)
        lm = eng.read_lm(w)
        print(f"{form}, window {w} ({len(sel)} far factors): ATE vs oracle {a:.3e} m, rot {r:.3e} rad, cost {lm['cost']:.9e} vs {costs[-1]:.9e}")
        assert a <= 1e-6 and r <= 1e-6 and lm["solve_failures"] == 0
        assert abs(lm["cost"] - costs[-1]) <= 1e-8 * abs(costs[-1])
        outs.append(got)
    # the far factors do change the answer (a loop closure pulls the far end of the window by millimetres)
    assert helpers.ate(outs[0], outs[1])[0] > 1e-4 and helpers.ate(outs[2], outs[1])[0] > 1e-4
    # a far factor whose older keyframe has left the window stops contributing: the window [30, n) knows only (40, 46)
    ref = Engine(EngineOpts(windows=1, capacity=n, **opts))
    helpers.load_engine(ref, 0, prob)
    ref.set_extra_between(0, fa[:1], fb[:1], far_rec[:1])
    for e, w in ((eng, 0), (ref, 0)):
        e.set_states(w, 0, prob["states"])
        e.set_prior(w, 30, synth.prior_record(prob["states"][30], np.array([1e-3] * 15)))
        e.set_range(w, 30, n)
        e.iterate(8)
    np.testing.assert_array_equal(eng.get_states(0, 30, n - 30), ref.get_states(0, 30, n - 30))
    # clearing the list gives the band-only engine back
    eng.set_extra_between(0, [], [], np.zeros((0, 28)))
    eng.set_extra_between(2, [], [], np.zeros((0, 28)))
    for w in (0, 1):
        eng.set_states(w, 0, prob["states"])
        eng.set_prior(w, 0, prob["prior"])
        eng.set_range(w, 0, n)
    eng.iterate(10)
    np.testing.assert_array_equal(eng.get_states(0, 0, n), eng.get_states(1, 0, n))
    with pytest.raises(VilFusionError) as ei:
        eng.set_extra_between(0, np.arange(9), np.arange(9) + 10, np.tile(far_rec[0], (9, 1)))
    assert ei.value.code == -6
    eng.close()
    ref.close()


@pytest.mark.parametrize("compat", [False, True])
def test_graph_manager_takes_any_pair_of_keys(oracle, compat):
    """vf_add_between with a span-6 factor and a loop-closure pair (GraphManager.cpp:83-88): accepted, solved, and the
    whole trajectory equals the oracle's on the same graph -- LM to convergence and the reference-compat one-update form
    (which goes through the same stages)."""
    from tests.test_gpu_graph_manager import _feed
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    n = 120
    seq = synth.make_sequence(33, n)
    far = [(30, 36), (10, 60), (11, 61)]
    rng = np.random.default_rng(6)
    recs = [_far_record(seq, a, b, rng) for a, b in far]
    gm = GraphManager(capacity=128, iterations=25, rel_tol=0, abs_tol=0, reference_compat=compat, max_far_factors=8)
    gm.setInitialState(seq.gt_states[0])
    gm.addIMUMeasurement(0.0, seq.imu_steps[0, 1:4], seq.imu_steps[0, 4:7])
    _feed(gm, seq, n)
    staged = gm.graphSize()
    for (a, b), r in zip(far, recs):
        gm.addBetweenFactor(a, b, (r[0:4], r[4:7]), np.eye(6) * 0.05)
    assert gm.graphSize() == staged + 3                       # they count as staged factors like any other
    for _ in range(6 if compat else 1):                       # (one iSAM2-like update per solve in compat mode: repeat to converge)
        gm.solve()
    assert gm.graphSize() == 0
    xs = gm.trajectory(0, n)
    imu = np.zeros((n, 190))
    for k in range(1, n):
        imu[k] = gm.imuFactor(k)
    g = np.array([0, 0, -9.81])
    states = np.zeros((n, 16)); states[0] = seq.gt_states[0]
    for k in range(1, n):
        states[k] = oracle.predict(imu[k], g, states[k - 1])
    m = seq.btw_a >= 1
    fa, fb = np.array([a for a, _ in far], dtype=np.int32), np.array([b for _, b in far], dtype=np.int32)
    prob = dict(n=n, states=states, imu=imu, btw_a=np.concatenate([seq.btw_a[m], fa]).astype(np.int32),
                btw_b=np.concatenate([seq.btw_b[m], fb]).astype(np.int32), btw=np.vstack([synth.between_records(seq)[m], np.array(recs)]),
                prior=synth.prior_record(states[0], REFERENCE_PRIOR_SIGMAS), gravity=g)
    win = helpers.oracle_window(oracle, prob)
    win.lm(iterations=40)
    ate, rot = helpers.ate(xs, win.states)
    print(f"GraphManager with far factors (reference_compat={compat}): ATE vs oracle {ate:.3e} m, rot {rot:.3e} rad")
    assert ate <= 1e-6 and rot <= 1e-6
    # the list is bounded (a handle made for eight; the default is 32: tests/test_gpu_far_capacity.py): a ninth far factor is refused like the band refused a wide one before
    for i in range(5):
        gm.addBetweenFactor(40 + i, 60 + i, (recs[0][0:4], recs[0][4:7]), np.eye(6))
    with pytest.raises(VilFusionError) as ei:
        gm.addBetweenFactor(50, 70, (recs[0][0:4], recs[0][4:7]), np.eye(6))
    assert ei.value.code == -6
    gm.close()


@pytest.mark.parametrize("lag", [0, 40])
def test_far_factors_survive_compaction_and_growth(lag):
    """The far list lives in window-local slots on the device and in absolute keys in the GraphManager: a handle whose engine
    compacts (fixed lag, 64 slots) or grows (whole history, 64 initial slots) while far factors are alive must publish what a
    roomy handle publishes; in fixed-lag mode the factors also age out of the window one after the other."""
    from tests.test_gpu_graph_manager import _stream
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    n = 230
    seq = synth.make_sequence(74, n)
    traj_t, acc, gyr = _stream(seq)
    rng = np.random.default_rng(9)
    # a span-8 factor every 14 keyframes (they age out of the 40-keyframe lag); whole history: every 28, i.e. VF_MAX_EXTRA = 8 in all
    far = {k: _far_record(seq, k - 8, k, rng) for k in range(20, n, 14 if lag else 28)}
    assert lag or len(far) == 8
    outs = []
    for cap in (128 if lag else 64, 512):       # (fixed lag: slots are reclaimed in whole tiles of 64 below the window)
        gm = GraphManager(capacity=cap, iterations=4, lag=lag, rel_tol=0, abs_tol=0)
        gm.setInitialState(seq.gt_states[0])
        out, i_imu = [], 0
        for k in range(1, n):
            while i_imu < traj_t.size and traj_t[i_imu] <= seq.kf_time[k] + 0.01:
                gm.addIMUMeasurement(traj_t[i_imu], acc[i_imu], gyr[i_imu]); i_imu += 1
            gm.reserveNode(seq.kf_time[k])
            for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov):
                if b == k and a >= 1:
                    gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
            if k in far:
                gm.addBetweenFactor(k - 8, k, (far[k][0:4], far[k][4:7]), np.eye(6) * 0.05)
            if k % 2 == 0:
                gm.solve()
                (q, t), v, b = gm.getState()
                out.append(np.concatenate([q, t, v, b]))
        outs.append(np.array(out))
        gm.close()
    small, big = outs
    assert small.shape == big.shape and np.isfinite(small).all()
    d = np.abs(small - big).max()
    print(f"lag {lag}: small handle (compacting / growing with far factors alive) vs a 512-slot one: largest difference {d:.3e}")
    assert d <= 1e-8


def test_far_factors_under_hip_graph_replay():
    """the Woodbury solve (6 extra band solves per slot in use, memsets, the combine kernel) is part of the launch sequence a
    use_hip_graph engine captures: replayed, it gives the bits of plain launches, and changing the list re-captures"""
    n = 120
    seq = synth.make_sequence(seed=93, n_kf=n)
    rng = np.random.default_rng(3)
    recs = np.array([_far_record(seq, 10, 100, rng), _far_record(seq, 30, 37, rng)])
    from tests.test_gpu_ingest import _engine
    eager, graph = _engine(None, [seq], n, 0), _engine(None, [seq], n, 0, use_hip_graph=True)
    for e in (eager, graph):
        e.set_extra_between(0, [10], [100], recs[:1])
        e.iterate(6)
        e.iterate(6)
    np.testing.assert_array_equal(eager.get_states(0, 0, n), graph.get_states(0, 0, n))
    for e in (eager, graph):
        e.set_extra_between(0, [10, 30], [100, 37], recs)
        e.iterate(6)
    np.testing.assert_array_equal(eager.get_states(0, 0, n), graph.get_states(0, 0, n))
    assert eager.read_lm(0) == graph.read_lm(0)
    enabled, captures, replays = graph.graph_info()
    assert enabled and replays == 3 and captures == 2
    eager.close()
    graph.close()


@pytest.mark.parametrize("cov", [1e-6, 10.0])
def test_far_factor_information_extremes(oracle, cov):
    """The Woodbury correction loses digits when U^T A^-1 U dwarfs the identity (a loop closure a million times stronger than
    the odometry) -- but an inexact step only slows the iteration down, the fixed point is g = 0 whatever the solve's
    accuracy: a very strong and a very weak far factor both end at the oracle's optimum."""
    n = 160
    seq = synth.make_sequence(seed=95, n_kf=n)
    prob = helpers.build_problem(oracle, seq, perturb=0.002)
    rng = np.random.default_rng(8)
    rec = _far_record(seq, 15, 65, rng, cov=cov, noise=(1e-4, 1e-3))[None]
    eng = Engine(EngineOpts(windows=1, capacity=n))
    helpers.load_engine(eng, 0, prob)
    eng.set_extra_between(0, [15], [65], rec)
    eng.iterate(40)
    p = dict(prob, btw_a=np.concatenate([prob["btw_a"], [15]]).astype(np.int32), btw_b=np.concatenate([prob["btw_b"], [65]]).astype(np.int32),
             btw=np.vstack([prob["btw"], rec]))
    win = helpers.oracle_window(oracle, p)
    costs, _, _ = win.lm(iterations=40)
    a, r = helpers.ate(eng.get_states(0, 0, n), win.states)
    lm = eng.read_lm(0)
    print(f"far factor with covariance {cov:g}: ATE vs oracle {a:.3e} m, cost {lm['cost']:.9e} vs {costs[-1]:.9e}, accepted {lm['accepted']} rejected {lm['rejected']}")
    assert a <= 1e-6 and r <= 1e-6 and lm["solve_failures"] == 0 and abs(lm["cost"] - costs[-1]) <= 1e-7 * abs(costs[-1])
    eng.close()


@pytest.mark.parametrize("closures", [((6, 72),), ((6, 72), (11, 75), (11, 64))])
def test_a_loop_closure_outlives_its_anchor_keyframe(oracle, closures):
    """iSAM2 keeps every BetweenFactor for good (GraphManager.cpp:83-88).  Here a far factor whose older keyframe is
    marginalised is marginalised WITH it (k_marginalize): it becomes a linear far factor over the marginal prior's three
    keyframes and its far end -- exact at the current linearisation --, every later marginalisation re-expresses it the same
    way, and once its far end is within the prior's reach it is absorbed into the prior: the information outlives both its
    ends.  An 80-keyframe window slides over a 170-keyframe clip with precise loop closures; the window is compared with the
    WHOLE-HISTORY batch optimum of the oracle (all keyframes, all factors) after 50 slides (the anchors long gone, the end
    keys still inside) and after 90 (both ends gone, the factors absorbed).  Dropping a factor when its anchor leaves (the
    behaviour until round 4, emulated) loses the closure entirely: the window ends exactly as far from the batch optimum as
    the closures move it.  Marginalised, the window follows the batch to 1e-7 m -- as close as a window with no far factor at
    all follows its own batch (round 5's first form, re-anchoring the factor on the next keyframe with the step between
    taken as exact, kept two thirds of the effect).  Several closures alive at once, two of them leaving with the same
    keyframe: marginalising a keyframe they all touch couples their far ends, so the window holds them as ONE linear factor
    with six rows per far end (marginalised one by one, each blind to the others, the same run ends 7e-3 m from the batch)."""
    total, n, K = 170, 80, 6
    seq = synth.make_sequence(seed=93, n_kf=total)
    prob = helpers.build_problem(oracle, seq)
    rng = np.random.default_rng(7)
    fa, fb = np.array([c[0] for c in closures], dtype=np.int32), np.array([c[1] for c in closures], dtype=np.int32)
    far = np.stack([_far_record(seq, a, b, rng, cov=1e-4, noise=(1e-4, 1e-3)) for a, b in closures])
    p_far = dict(prob, btw_a=np.concatenate([prob["btw_a"], fa]).astype(np.int32), btw_b=np.concatenate([prob["btw_b"], fb]).astype(np.int32),
                 btw=np.vstack([prob["btw"], far]))
    CHECK = (50, total - n)                              # slides after which the window is compared
    refs = {}
    for name, p in (("with", p_far), ("without", prob)):
        for s in CHECK:                                  # the batch optimum of the history up to the window's end
            win = helpers.oracle_window(oracle, p, 0, s + n)
            win.lm(iterations=40)
            refs[name, s] = win.states[s:s + n].copy()

    def run(mode):
        eng = Engine(EngineOpts(windows=1, capacity=total))
        helpers.load_engine(eng, 0, prob, 0, n)
        if mode != "none":
            eng.set_extra_between(0, fa, fb, far)
        eng.iterate(40)
        out = {}
        for s in range(1, total - n + 1):
            if mode == "dropped":                        # (what the library did until round 4: a factor is gone once its anchor has left)
                keep = fa >= s
                eng.set_extra_between(0, fa[keep], fb[keep], far[keep])
            eng.slide(marginalize=True)
            eng.iterate(K)
            if s in CHECK:
                out[s] = (eng.get_states(0, s, n), eng.get_extra_between(0), eng.get_linear_far(0))
        lm = eng.read_lm(0)
        eng.close()
        return out, lm

    tr, lm = run("transport")
    dr, _ = run("dropped")
    no, _ = run("none")
    for s in CHECK:
        moved = helpers.ate(refs["with", s], refs["without", s])[0]
        e_t = helpers.ate(tr[s][0], refs["with", s])[0]
        e_d = helpers.ate(dr[s][0], refs["with", s])[0]
        e_n = helpers.ate(no[s][0], refs["without", s])[0]
        ea, eb, _, transported, ended, absorbed = tr[s][1]
        linear = tr[s][2].tolist()
        print(f"loop closures {closures}, {n}-keyframe window after {s} slides: they move this window's batch optimum by {moved:.3e} m; fixed lag vs "
              f"whole-history batch: marginalised with their anchors {e_t:.3e} m, dropped at the anchor's exit {e_d:.3e} m (no far factor at all, vs its own batch: "
              f"{e_n:.3e} m); far list a = {ea.tolist()} b = {eb.tolist()}, linear far factors ending at {linear}, made linear {transported}, absorbed {absorbed}")
        assert moved > 1e-3 and abs(e_d - moved) < 0.05 * moved and e_t < 1e-6 and e_n < 1e-5
        assert ea.tolist() == [] and transported == len(closures) and ended == 0
        if s == 50:
            assert sorted(linear) == sorted(b for _, b in closures) and absorbed == 0
        else:
            assert linear == [] and absorbed == len(closures)
    assert lm["solve_failures"] == 0


def test_graph_manager_keeps_a_loop_closure_across_its_lag():
    """The same through the GraphManager (vf_add_between routes the wide factor to the far list; vf_solve marginalises with a
    lag of 40): the published estimate after the anchor key has left the window follows the whole-history handle (lag = 0,
    the reference's unbounded graph) fed the same factors, several times closer than a lag-40 handle that never got the
    loop closure -- to 2e-7 m, where the handle that never got the closure is 1.4e-2 m away."""
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    total, a0, b0 = 110, 5, 38            # lag 40: key 5 leaves at solve 46, key 38 at solve 79; the last 30 keys hold neither
    seq = synth.make_sequence(seed=94, n_kf=total, keep_raw=True)
    rng = np.random.default_rng(8)
    rec = _far_record(seq, a0, b0, rng, cov=1e-4, noise=(1e-4, 1e-3))
    cov = {c: np.eye(6) * c for c in (synth.VIO_COV, synth.LIDAR_COV)}
    by_end = {int(b): i for i, b in enumerate(seq.btw_b)}

    def run(lag, closure):
        gm = GraphManager(capacity=256, lag=lag, iterations=6, rel_tol=0.0, abs_tol=0.0)
        gm.setInitialState(seq.gt_states[0])
        gm.addIMUMeasurement(0.0, seq.imu_acc[0], seq.imu_gyro[0])
        i_imu = 0
        for k in range(1, total):
            while i_imu < seq.imu_t.size and seq.imu_t[i_imu] <= seq.kf_time[k] + 0.011:
                gm.addIMUMeasurement(seq.imu_t[i_imu], seq.imu_acc[i_imu], seq.imu_gyro[i_imu])
                i_imu += 1
            assert gm.reserveNode(seq.kf_time[k]) == k
            if k in by_end:
                i = by_end[k]
                gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), cov[float(seq.btw_cov[i])])
            if closure and k == b0:
                gm.addBetweenFactor(a0, b0, (rec[0:4], rec[4:7]), np.eye(6) * 1e-4)
            gm.solve()
        st = gm.lmStats()
        tr = gm.trajectory(total - 31, 30)             # the last 30 keys: inside every handle's window
        gm.close()
        return tr, st

    whole, st0 = run(0, True)
    lagged, st1 = run(40, True)
    blind, st2 = run(40, False)
    d_keep = helpers.ate(lagged, whole)[0]
    d_blind = helpers.ate(blind, whole)[0]
    print(f"GraphManager, loop closure ({a0}, {b0}), lag 40 vs whole history over the last 30 keys: with the closure marginalised with its anchor {d_keep:.3e} m, never given the closure {d_blind:.3e} m")
    assert st1["solve_failures"] == 0 and d_keep < 2e-6 and d_blind > 1e-3


def test_far_capacity_counts_the_factors_the_engine_has_taken_over():
    """max_far_factors (here 8 = VF_MAX_EXTRA, what the LDS forms hold; a handle's default is 32) bounds the far factors ALIVE in a window: the ones in the GraphManager's list and the far ends of the
    engine's linear far factor (far factors already marginalised with their older key) together.  A ninth is refused where it
    is added (VF_ERR_CAPACITY from vf_add_between, as before) -- never by a later vf_solve -- and there is room again once a
    far end has been folded into the marginal prior."""
    from tests.test_gpu_graph_manager import _stream
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    n, lag = 150, 40
    seq = synth.make_sequence(77, n)
    traj_t, acc, gyr = _stream(seq)
    rng = np.random.default_rng(11)
    gm = GraphManager(capacity=256, iterations=4, lag=lag, rel_tol=0, abs_tol=0, max_far_factors=8)
    gm.setInitialState(seq.gt_states[0])
    i_imu, taken, refused = 0, [], []
    for k in range(1, n):
        while i_imu < traj_t.size and traj_t[i_imu] <= seq.kf_time[k] + 0.01:
            gm.addIMUMeasurement(traj_t[i_imu], acc[i_imu], gyr[i_imu]); i_imu += 1
        gm.reserveNode(seq.kf_time[k])
        for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov):
            if b == k and a >= 1:
                gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
        if k >= 45 and k % 2 == 1 and k < 120:            # a closure (k - 36, k): its older key leaves the lag 4 solves later
            rec = _far_record(seq, k - 36, k, rng)
            try:
                gm.addBetweenFactor(k - 36, k, (rec[0:4], rec[4:7]), np.eye(6) * 0.05)
                taken.append(k)
            except VilFusionError as exc:
                assert exc.code == -6, exc
                refused.append(k)
        gm.solve()                                          # (raises if a solve fails)
    st = gm.lmStats()
    (q, t), v, b = gm.getState()
    gm.close()
    print(f"closures taken at keys {taken}, refused (capacity) at {refused}; lm {st}")
    # a closure added at key k is alive until key k + 37 has been added (its far end k within 3 keys of the window's head):
    # one every 2 keys fills the 8 slots after 16 keys, then one is taken whenever one has been folded into the prior
    assert taken[:8] == list(range(45, 61, 2)) and refused and refused[0] == 61 and len(taken) > 10
    assert st["solve_failures"] == 0 and np.isfinite(t).all()


def test_refined_windows_take_far_factors_as_rows_of_the_operator(oracle):
    """A window long enough to be refined (vf_engine_opts.refine_iterations; here forced on a short one) that holds far
    factors -- nonlinear ones and, after their anchors have been marginalised, the linear far factor -- has their rows in the
    operator of the refinement's conjugate gradients (k_far_apply: six more rows of J per slot) and the Woodbury solve as its
    preconditioner.  Same normal equations as the unrefined engine's: the two stay together over 60 marginalised slides,
    through the conversion of both closures and the folding of both into the prior."""
    total, n, K = 130, 60, 6
    seq = synth.make_sequence(seed=95, n_kf=total)
    prob = helpers.build_problem(oracle, seq)
    rng = np.random.default_rng(12)
    closures = ((4, 50), (9, 57))
    fa, fb = np.array([c[0] for c in closures], dtype=np.int32), np.array([c[1] for c in closures], dtype=np.int32)
    far = np.stack([_far_record(seq, a, b, rng, cov=1e-4, noise=(1e-4, 1e-3)) for a, b in closures])
    out = {}
    for name, opts in (("woodbury", {}), ("refined", dict(refine_iterations=8, lm_excursion=0))):
        eng = Engine(EngineOpts(windows=1, capacity=total, **opts))
        helpers.load_engine(eng, 0, prob, 0, n)
        eng.set_extra_between(0, fa, fb, far)
        eng.iterate(30)
        snaps = []
        for s in range(1, 61):
            eng.slide(marginalize=True)
            eng.iterate(K)
            if s in (3, 8, 30, 49, 60):               # both nonlinear / one converted / both linear / first folded / both folded
                snaps.append((eng.get_states(0, s, n), eng.get_extra_between(0)[0].tolist(), eng.get_linear_far(0).tolist()))
        out[name] = (snaps, eng.read_lm(0))
        eng.close()
    for (sa, na, la), (sb, nb, lb) in zip(out["woodbury"][0], out["refined"][0]):
        d = np.abs(sa - sb).max()
        print(f"nonlinear far list {na}, linear far ends {la}: Woodbury vs refined operator, largest state difference {d:.3e}")
        assert na == nb and la == lb and d <= 1e-8
    assert [x[1:] for x in out["woodbury"][0]] == [([4, 9], []), ([9], [50]), ([], [50, 57]), ([], [57]), ([], [])]
    assert out["woodbury"][1]["solve_failures"] == 0 and out["refined"][1]["solve_failures"] == 0


@pytest.mark.parametrize("tol", [0.0, None])
def test_random_loop_closures_through_a_fixed_lag_handle(tol):
    """Bookkeeping under load: a GraphManager with a lag of 50 and 128 slots (so that it compacts while far factors of both
    kinds are alive; all made for eight far factors, the capacity of the LDS forms) takes a loop closure between random keys every few keyframes -- up to the capacity of eight alive at a
    time; the ones refused are simply not added to any handle -- and must publish what (a) a roomy handle with the same lag
    publishes (to 1e-8: compaction moves slots, nothing else) and (b) a whole-history handle publishes (lag = 0: every factor
    kept for good, as in the reference) to 1e-6 m: far factors converted, re-expressed at every marginalisation, sharing their
    older keys, folded into the prior one after the other.  tol = None: the default termination rule, under which a window
    still carries its `done` flag when the next solve marginalises -- the far factors must be linearised for that all the same
    (a compaction in between has zeroed their records: tools/far_soak.py found the marginal's far-end block singular there)."""
    from tests.test_gpu_graph_manager import _stream
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    n, lag = 260, 50
    seq = synth.make_sequence(79, n)
    traj_t, acc, gyr = _stream(seq)
    rng = np.random.default_rng(21)
    plan = {}
    k = 30
    while k < n - 5:
        span = int(rng.integers(8, min(lag - 6, k - 1)))
        plan[k] = (k - span, _far_record(seq, k - span, k, rng, cov=1e-3, noise=(3e-4, 3e-3)))
        k += int(rng.integers(3, 12))
    handles = {"small": GraphManager(capacity=128, iterations=5, lag=lag, rel_tol=tol, abs_tol=tol, max_far_factors=8),
               "roomy": GraphManager(capacity=512, iterations=5, lag=lag, rel_tol=tol, abs_tol=tol, max_far_factors=8),
               "whole": GraphManager(capacity=512, iterations=5, lag=0, rel_tol=tol, abs_tol=tol, max_far_factors=8)}
    out = {name: [] for name in handles}
    for gm in handles.values():
        gm.setInitialState(seq.gt_states[0])
    i_imu, taken, refused = 0, 0, 0
    for k in range(1, n):
        j = i_imu
        for name, gm in handles.items():
            j = i_imu
            while j < traj_t.size and traj_t[j] <= seq.kf_time[k] + 0.01:
                gm.addIMUMeasurement(traj_t[j], acc[j], gyr[j]); j += 1
            gm.reserveNode(seq.kf_time[k])
            for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov):
                if b == k and a >= 1:
                    gm.addBetweenFactor(int(a), int(b), (q, t), np.eye(6) * c)
        i_imu = j
        if k in plan:
            a, rec = plan[k]
            try:                                          # the small handle decides (the fixed-lag ones agree; the whole-history one has room when they do)
                handles["small"].addBetweenFactor(a, k, (rec[0:4], rec[4:7]), np.eye(6) * 1e-3)
                ok = True
            except VilFusionError as exc:
                assert exc.code == -6, exc
                ok = False
            if ok:
                taken += 1
                handles["roomy"].addBetweenFactor(a, k, (rec[0:4], rec[4:7]), np.eye(6) * 1e-3)
                try:
                    if "whole" in handles:
                        handles["whole"].addBetweenFactor(a, k, (rec[0:4], rec[4:7]), np.eye(6) * 1e-3)
                except VilFusionError as exc:             # (the whole-history handle never lets a far factor go: its eight slots fill up)
                    assert exc.code == -6, exc
                    handles["whole"] = None
            else:
                refused += 1
        for name, gm in handles.items():
            if gm is None:
                continue
            gm.solve()
            (q, t), v, b = gm.getState()
            out[name].append(np.concatenate([q, t, v, b]))
        if "whole" in handles and handles["whole"] is None:
            handles.pop("whole")
    stats = {name: gm.lmStats() for name, gm in handles.items()}
    for gm in handles.values():
        gm.close()
    small, roomy, whole = np.array(out["small"]), np.array(out["roomy"]), np.array(out["whole"])
    d_room = np.abs(small - roomy).max()
    m = whole.shape[0]
    d_whole = np.sqrt(np.mean(np.sum((small[:m, 4:7] - whole[:, 4:7]) ** 2, axis=1))) if m else float("nan")
    print(f"{taken} closures taken, {refused} refused for capacity; small vs roomy handle: {d_room:.3e}; fixed lag vs whole history over the first {m} solves "
          f"(the whole-history handle holds at most 8 far factors for good): position rms {d_whole:.3e} m; lm {stats}")
    assert taken >= 12 and d_room <= 1e-8 and m >= 60 and d_whole <= (1e-6 if tol == 0.0 else 1e-4)      # (windows that stop at the rule's 1e-5 are that far from converged)
    assert all(s["solve_failures"] == 0 for s in stats.values())


def test_batched_woodbury_columns_are_the_sequential_ones_bit_for_bit(oracle):
    """vf_engine_opts.far_batch_columns: a single-window engine (the GraphManager's) solves the 6 Woodbury columns of every far
    factor as ONE partitioned solve of an internal 48-window engine -- window q a copy of the window's H with column q of U as
    right-hand side -- where a batch engine runs its band solver once per column.  The same kernels on the same numbers:
    states, costs and trial counts agree to the last bit, nonlinear far factors and (after marginalised slides) linear ones."""
    total, n = 120, 70
    seq = synth.make_sequence(seed=97, n_kf=total)
    prob = helpers.build_problem(oracle, seq, perturb=0.002)
    rng = np.random.default_rng(13)
    closures = ((3, 60), (8, 41), (8, 66), (10, 50), (12, 64), (15, 58), (20, 69), (25, 45))       # (VF_MAX_EXTRA of them: 48 columns)
    fa, fb = np.array([c[0] for c in closures], dtype=np.int32), np.array([c[1] for c in closures], dtype=np.int32)
    far = np.stack([_far_record(seq, a, b, rng, cov=1e-3) for a, b in closures])
    out = []
    for batch in (1, 0):
        eng = Engine(EngineOpts(windows=1, capacity=total, far_batch_columns=batch))
        helpers.load_engine(eng, 0, prob, 0, n)
        eng.set_extra_between(0, fa, fb, far)
        eng.iterate(12)
        snaps = [(eng.get_states(0, 0, n), eng.read_lm(0))]
        for s in range(1, 41):
            eng.slide(marginalize=True)
            eng.iterate(4)
            if s in (5, 12, 40):
                snaps.append((eng.get_states(0, s, n), eng.read_lm(0), eng.get_linear_far(0).tolist()))
        out.append(snaps)
        eng.close()
    for a, b in zip(*out):
        np.testing.assert_array_equal(a[0], b[0])
        assert a[1:] == b[1:]
    assert out[0][2][2] == [60, 41, 66, 50] and out[0][3][2] == [60, 66, 50, 64, 58, 69, 45] and out[0][0][1]["solve_failures"] == 0 and out[0][3][1]["solve_failures"] == 0


def test_windows_of_a_batch_engine_keep_their_own_far_factors(oracle):
    """A batch engine whose windows hold different far factors (none, one, three sharing a key) through 50 marginalised slides
    -- conversions, re-expressions and foldings at different slides in different windows, the Woodbury columns solved one
    after the other for the whole batch -- against one single-window engine per window (whose columns go through the column
    engine): the same states to the last bit."""
    total, n, K = 120, 60, 4
    seqs = [synth.make_sequence(seed=300 + w, n_kf=total) for w in range(4)]
    probs = [helpers.build_problem(oracle, s, perturb=0.002) for s in seqs]
    rng = np.random.default_rng(17)
    plans = [(), ((5, 48),), ((2, 40), (9, 55), (9, 31)), ((20, 58), (1, 12))]
    fars = [(np.array([c[0] for c in p], dtype=np.int32), np.array([c[1] for c in p], dtype=np.int32),
             np.stack([_far_record(seqs[w], a, b, rng, cov=1e-3) for a, b in p]) if p else np.zeros((0, 28))) for w, p in enumerate(plans)]

    def run(eng, windows):
        for slot, w in enumerate(windows):
            helpers.load_engine(eng, slot, probs[w], 0, n)
            if len(fars[w][0]):
                eng.set_extra_between(slot, *fars[w])
        eng.iterate(15)
        out = []
        for s in range(1, 51):
            eng.slide(marginalize=True)
            eng.iterate(K)
            if s in (4, 12, 30, 50):
                out.append([(eng.get_states(slot, s, n), eng.get_linear_far(slot).tolist()) for slot in range(len(windows))])
        lm = [eng.read_lm(slot) for slot in range(len(windows))]
        eng.close()
        return out, lm

    batch, lm_b = run(Engine(EngineOpts(windows=4, capacity=total)), [0, 1, 2, 3])
    for w in range(4):
        single, lm_s = run(Engine(EngineOpts(windows=1, capacity=total)), [w])
        for snap_b, snap_s in zip(batch, single):
            np.testing.assert_array_equal(snap_b[w][0], snap_s[0][0])
            assert snap_b[w][1] == snap_s[0][1]
        assert lm_b[w] == lm_s[0] and lm_b[w]["solve_failures"] == 0
    assert [x[1] for x in batch[1]] == [[], [48], [40, 55, 31], []]            # after 12 slides: every anchor but 20 has left, (1, 12) has been folded
    assert [x[1] for x in batch[2]] == [[], [48], [40, 55], [58]] and [x[1] for x in batch[3]] == [[], [], [55], [58]]
