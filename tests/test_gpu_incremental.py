"""-m gpu: incremental updates (vf_engine_opts.incremental, vf_graph_opts.incremental) -- the banded form of what
ISAM2::update does with relinearizeThreshold 1e-4 / relinearizeSkip 1 (GraphManager.cpp:37-43,126-127): only the keyframes from
the first one that moved, or whose factors are new, are linearised, assembled and eliminated again (the forward sweep restarts
from a checkpoint of its trailing window), and the back substitution stops once the increments come out as they were.
Checked against the same update done over the whole window: to the bit where the whole-window solve is the one-wave sweep
(same arithmetic in the same order), to 1e-9 m where it is the partitioned form."""
import numpy as np
import pytest

from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth

pytestmark = pytest.mark.gpu

SWEEP = dict(chunks=1, sweep_two_sided_max=0, solve_assemble_min=0, refine_iterations=0, lm_excursion=0)


def test_growing_window_equals_the_whole_window_sweep_to_the_bit(oracle):
    """one window that grows by 1-3 keyframes per update (the GraphManager's whole-history mode), 150 updates"""
    n0, n = 40, 330
    seq = synth.make_sequence(seed=91, n_kf=n)
    prob = helpers.build_problem(oracle, seq, perturb=0.003)
    full = Engine(EngineOpts(windows=1, capacity=n + 8, **SWEEP))
    inc = Engine(EngineOpts(windows=1, capacity=n + 8, incremental=1, **SWEEP))
    for e in (full, inc):
        helpers.load_engine(e, 0, prob, 0, n0)
    hi, rng, partial, spans = n0, np.random.default_rng(5), 0, []
    for u in range(150):
        full.isam_step(1e-4)
        inc.isam_step(1e-4)
        np.testing.assert_array_equal(inc.get_states(0, 0, hi), full.get_states(0, 0, hi))        # linearisation points
        np.testing.assert_array_equal(inc.get_estimate(0, 0, hi), full.get_estimate(0, 0, hi))    # theta (+) delta
        np.testing.assert_array_equal(inc.read_delta(0, 0, hi), full.read_delta(0, 0, hi))
        info = inc.incremental_info(0)
        if u > 0:
            partial += info["first_eliminated"] > 0
            spans.append((hi - info["first_eliminated"], hi - info["last_substituted"]))
        g = int(rng.integers(1, 4))
        if hi + g > n:
            break
        hi += g
        for e in (full, inc):
            e.set_range(0, 0, hi)
    info = inc.incremental_info(0)
    print(f"{info['updates']} incremental updates, {info['whole_window_updates']} over the whole window; keyframes eliminated / substituted "
          f"again per update: median {np.median([s[0] for s in spans]):.0f} / {np.median([s[1] for s in spans]):.0f}, max "
          f"{max(s[0] for s in spans)} / {max(s[1] for s in spans)}, final window {hi}")
    assert info["whole_window_updates"] == 1 and partial >= 0.8 * len(spans)
    # (every keyframe here starts 3 mm / 3 mrad off, thirty thresholds: the updates relinearise far back while the window converges)
    full.close(); inc.close()


def test_an_entry_point_the_bookkeeping_does_not_follow_voids_the_factorisation(oracle):
    n = 96
    seq = synth.make_sequence(seed=92, n_kf=n)
    prob = helpers.build_problem(oracle, seq, perturb=0.003)
    full = Engine(EngineOpts(windows=1, capacity=n + 8, **SWEEP))
    inc = Engine(EngineOpts(windows=1, capacity=n + 8, incremental=1, **SWEEP))
    for e in (full, inc):
        helpers.load_engine(e, 0, prob, 0, n)
        e.isam_step(1e-4)
        e.isam_step(1e-4)
    assert inc.incremental_info(0)["whole_window_updates"] == 1
    moved = prob["states"][30:32].copy()
    moved[:, 4:7] += 0.02
    for e in (full, inc):
        e.set_states(0, 30, moved)           # keyframes in the middle of the window rewritten behind the engine's back
        e.isam_step(1e-4)
    assert inc.incremental_info(0)["whole_window_updates"] == 2 and inc.incremental_info(0)["first_eliminated"] == 0
    np.testing.assert_array_equal(inc.get_estimate(0, 0, n), full.get_estimate(0, 0, n))
    # ... and a between factor that arrives late, for a keyframe well inside the window, is followed (one-window engines)
    a, b = 50, 52
    Ra, Rb = synth.quat_to_rot(seq.gt_states[a, :4]), synth.quat_to_rot(seq.gt_states[b, :4])
    rec = np.zeros(28)
    rec[:4] = synth.rot_to_quat(Ra.T @ Rb)
    rec[4:7] = Ra.T @ (seq.gt_states[b, 4:7] - seq.gt_states[a, 4:7])
    rec[7:] = prob["btw"][0][7:]
    for e in (full, inc):
        e.clear_between(0, b, 1)
        e.set_between(0, np.array([a], dtype=np.int32), np.array([b], dtype=np.int32), rec.reshape(1, 28))
        e.isam_step(1e-4)
    info = inc.incremental_info(0)
    assert info["whole_window_updates"] == 3 or 0 < info["first_eliminated"] <= b - 6
    np.testing.assert_array_equal(inc.get_estimate(0, 0, n), full.get_estimate(0, 0, n))
    full.close(); inc.close()


def test_fixed_lag_batch_slides(oracle):
    """several windows, each slid by one keyframe per update (marginalise the oldest, append one): the incremental engine keeps
    the panels of an elimination that ran on from the very first window, i.e. the exact marginal at the linearisation points of
    the time, where the whole-window engine restarts from the marginal prior k_marginalize re-derives at every slide (with its
    gauge floor): equal to rounding plus what the floor adds, not to the bit"""
    B, n, updates = 3, 120, 60
    seqs = [synth.make_sequence(seed=95 + w, n_kf=n + updates + 2) for w in range(B)]
    engines = {}
    for name, extra in (("full", {}), ("inc", dict(incremental=1))):
        e = Engine(EngineOpts(windows=B, capacity=n + updates + 8, gauge_floor=0.0, **SWEEP, **extra))
        for w, seq in enumerate(seqs):
            prob = helpers.build_problem(oracle, seq, perturb=0.002)
            helpers.load_engine(e, w, prob, 0, n)
        for _ in range(6):
            e.isam_step(1e-4)
        engines[name] = e
    worst, started = 0.0, []
    for u in range(updates):
        for e in engines.values():
            e.slide(helpers.REFERENCE_PRIOR_SIGMAS, marginalize=True)
            e.isam_step(1e-4)
        for w in range(B):
            a = engines["inc"].get_estimate(w, u + 1, n)
            b = engines["full"].get_estimate(w, u + 1, n)
            worst = max(worst, helpers.ate(a, b)[0])
            started.append(engines["inc"].incremental_info(w)["first_eliminated"] - (u + 1))
    info = engines["inc"].incremental_info(0)
    print(f"fixed-lag slides, {B} windows x {updates} updates: worst ATE incremental vs whole-window {worst:.3e} m; the sweep started "
          f"{np.median(started):.0f} keyframes into the window (median; max {max(started)}; window {n}); whole-window updates {info['whole_window_updates']} of {info['updates']}")
    # (where the sweep starts is the stream's business -- on this one every update moves the whole window by more than the
    # threshold, DESIGN.md "Incremental updates" -- what is checked is that wherever it starts the result is the whole-window one)
    assert worst <= 1e-7 and info["whole_window_updates"] == 1
    for e in engines.values():
        e.close()


def test_graph_manager_incremental_equals_full_reelimination():
    """the drop-in surface: reference_compat with incremental = 1 and = 2 (whole history every time, same sweep) on the same
    stream, one solve per keyframe: bit for bit; and against the default handle, which solves by the partitioned form"""
    from vil_sensor_fusion_amd import VilFusionError
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    with pytest.raises(VilFusionError):
        GraphManager(capacity=64, incremental=True)                  # the incremental update is the iSAM2-like one
    n = 420
    seq = synth.make_sequence(seed=93, n_kf=n)

    def run(**kw):
        gm = GraphManager(capacity=128, reference_compat=True, **kw)      # (grows twice on the way: 128 -> 256 -> 512)
        gm.setInitialState(seq.gt_states[0])
        gm.addIMUMeasurement(0.0, seq.imu_steps[0, 1:4], seq.imu_steps[0, 4:7])
        t, out = 0.0, []
        for k in range(1, n):
            for s in seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]:
                t += s[0]
                gm.addIMUMeasurement(t, s[1:4], s[4:7])
            gm.reserveNode(t)
            for i in np.nonzero(seq.btw_b == k)[0]:
                gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), np.eye(6) * seq.btw_cov[i])
            gm.solve()
            (q, p), v, b = gm.getState()
            out.append(np.concatenate([q, p, v, b]))
        traj = gm.trajectory(0, n)
        gm.close()
        return np.array(out), traj
    pub_i, traj_i = run(incremental=True)
    pub_w, traj_w = run(incremental=2)            # the same kernels from the first keyframe of the history at every solve
    np.testing.assert_array_equal(pub_i, pub_w)
    np.testing.assert_array_equal(traj_i, traj_w)
    pub_f, traj_f = run()
    gap = np.linalg.norm(pub_i[:, 4:7] - pub_f[:, 4:7], axis=1)
    ate, rot = helpers.ate(traj_i, traj_f)
    print(f"GraphManager reference_compat over {n} solves: incremental = whole-history re-elimination by the same sweep, bit for bit; against the "
          f"default handle (partitioned solve: another elimination order, and a keyframe whose increment is within rounding of the threshold "
          f"relinearises in one and not in the other): published position gap max {gap.max():.3e} m; smoothed trajectory at the end ATE {ate:.3e} m, rot {rot:.3e} rad")
    assert gap.max() <= 2e-5 and ate <= 1e-6 and rot <= 1e-6
