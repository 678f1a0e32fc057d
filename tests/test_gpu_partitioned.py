"""-m gpu: the partitioned solve K4p (P chunks per window joined by 27-dof separators;
vf_engine_opts.chunks) against the one-sweep band solver and the CPU oracle.

Both forms are Cholesky factorisations of the same block-banded system in different elimination
orders (nested dissection vs. natural), so the increments agree to cond*eps, not bit for bit.
Gates: backward error of the GPU increment on the GPU's own system (extended precision) <= 1e-9,
no worse than 50x the oracle's scalar band Cholesky; LM trajectories: ATE <= 1e-6 m."""
import numpy as np
import pytest

from tests import helpers
from tests.test_gpu_parity import band_matvec, relerr
from vil_sensor_fusion_amd import synth

pytestmark = pytest.mark.gpu

N = 200


def make_engine(oracle, chunks, ranges, perturb=0.01, seeds=None):
    from vil_sensor_fusion_amd import Engine, EngineOpts
    eng = Engine(EngineOpts(windows=len(ranges), capacity=N + 8, chunks=chunks))
    probs = []
    for w, (lo, hi) in enumerate(ranges):
        seq = synth.make_sequence(seed=w if seeds is None else seeds[w], n_kf=N)
        prob = helpers.build_problem(oracle, seq, perturb=perturb)
        helpers.load_engine(eng, w, prob, lo=lo, hi=hi)
        probs.append(prob)
    return eng, probs


@pytest.mark.parametrize("chunks", [2, 3, 5, 16])
def test_partitioned_solve_matches_band_system(oracle, chunks):
    ranges = [(0, N), (5, 150), (0, 64), (10, 47)]
    eng, probs = make_engine(oracle, chunks, ranges)
    eng.linearize(0)
    eng.assemble()
    eng.solve()
    for w, (lo, hi) in enumerate(ranges):
        n = hi - lo
        H, g = eng.read_normal(w, lo, n)
        d = eng.read_delta(w, lo, n)
        rc, do = oracle.band_solve(H, g, 1e-5)
        assert rc == 0
        Hl, gl = H.astype(np.longdouble), g.astype(np.longdouble)
        scale = np.abs(gl).max()
        bg = float(np.abs(band_matvec(Hl, np.longdouble(1e-5), d.astype(np.longdouble)) + gl).max() / scale)
        bo = float(np.abs(band_matvec(Hl, np.longdouble(1e-5), do.astype(np.longdouble)) + gl).max() / scale)
        print(f"P={chunks} window {w} n={n}: backward error gpu {bg:.3e} oracle {bo:.3e}; forward diff {relerr(d, do):.3e}")
        assert bg < 1e-9
        assert bg < 50 * bo + 1e-13
        assert relerr(d, do) < 1e-3
        assert eng.read_lm(w)["solve_failures"] == 0
    eng.close()


def test_partitioned_equals_one_sweep_increment(oracle):
    ranges = [(0, N), (3, 171)]
    ref, _ = make_engine(oracle, 1, ranges)      # whole-window sweeps
    par, _ = make_engine(oracle, 7, ranges)
    for e in (ref, par):
        e.linearize(0)
        e.assemble()
        e.solve()
    for w, (lo, hi) in enumerate(ranges):
        a, b = ref.read_delta(w, lo, hi - lo), par.read_delta(w, lo, hi - lo)
        # two elimination orders of a system with cond ~ 1e11 (prior information 1e14 next to
        # between-factor information 1e1): agreement to cond * eps, same gate as against the oracle
        print("one-sweep vs partitioned increment, relative difference", relerr(b, a))
        assert 0 < relerr(b, a) < 1e-3
    ref.close()
    par.close()


@pytest.mark.parametrize("chunks", [4, 16])
def test_partitioned_lm_trajectory_parity(oracle, chunks):
    ranges = [(0, N), (0, 120)]
    eng, probs = make_engine(oracle, chunks, ranges)
    eng.iterate(6)
    for w, (lo, hi) in enumerate(ranges):
        win = helpers.oracle_window(oracle, probs[w], lo=lo, hi=hi)
        win.lm(iterations=6)
        est = eng.get_states(w, lo, hi - lo)
        a, _ = helpers.ate(est, win.states)
        lm = eng.read_lm(w)
        print(f"P={chunks} window {w}: ATE vs oracle {a:.3e} m, cost {lm['cost']:.6e} (oracle {win.cost():.6e}), accepted {lm['accepted']}")
        assert a <= 1e-6
        assert lm["solve_failures"] == 0
        assert abs(lm["cost"] - win.cost()) <= 1e-6 * max(win.cost(), 1e-12)
    eng.close()


def test_partitioned_short_windows_fall_back(oracle):
    """Windows too short for the requested chunk count use fewer chunks (or one whole-window sweep)."""
    ranges = [(0, 5), (0, 11), (2, 22), (0, 35), (0, 1)]
    eng, probs = make_engine(oracle, 16, ranges)
    eng.iterate(4)
    for w, (lo, hi) in enumerate(ranges):
        win = helpers.oracle_window(oracle, probs[w], lo=lo, hi=hi)
        win.lm(iterations=4)
        a, _ = helpers.ate(eng.get_states(w, lo, hi - lo), win.states)
        print(f"n={hi - lo}: ATE {a:.3e}")
        assert a <= 1e-6
        assert eng.read_lm(w)["solve_failures"] == 0
    eng.close()


def test_partitioned_with_marginal_prior(oracle):
    """Fixed-lag slides (marginal prior on the first three keyframes) with the partitioned solve
    follow the one-sweep engine."""
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    ranges = [(0, 100)]
    engines = []
    for chunks in (1, 5):
        eng, _ = make_engine(oracle, chunks, ranges, perturb=0.0)
        eng.iterate(3)
        engines.append(eng)
    for step in range(6):
        for eng in engines:
            eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
            eng.iterate(3)
    a = engines[0].get_states(0, 6, 100)
    b = engines[1].get_states(0, 6, 100)
    d, _ = helpers.ate(a, b)
    print("one-sweep vs partitioned after 6 marginalised slides: ATE", d)
    assert d <= 1e-6
    for e in engines:
        e.close()


def test_config3_batch_lm_with_degeneracy_metrics(oracle):
    """BASELINE.json configs[2]: a 1000-pose batch LM solve with the partitioned Cholesky (16 chunks,
    MFMA Schur / spike products), degeneracy detection on: K6 over the 1000 per-keyframe 6x6 pose
    information blocks of the converged window, against the numpy restatement of the reference's
    metric library."""
    from oracle import degeneracy_oracle as dor
    from vil_sensor_fusion_amd import Engine, EngineOpts, degeneracy as dg
    n = 1000
    seq = synth.make_sequence(seed=31, n_kf=n)
    prob = helpers.build_problem(oracle, seq, perturb=0.005)
    eng = Engine(EngineOpts(windows=1, capacity=n, chunks=16))
    helpers.load_engine(eng, 0, prob)
    eng.iterate(8)
    win = helpers.oracle_window(oracle, prob)
    win.lm(iterations=8)
    a, r = helpers.ate(eng.get_states(0, 0, n), win.states)
    lm = eng.read_lm(0)
    print(f"config 3: ATE vs oracle {a:.3e} m, cost {lm['cost']:.6e}, accepted {lm['accepted']}")
    assert a <= 1e-6 and r <= 1e-6 and lm["solve_failures"] == 0
    eng.linearize(0)
    eng.assemble()
    info = eng.pose_information(0, 1, n - 1)                    # keyframe 0 carries the 1e14 prior
    mats = info.transpose(2, 0, 1)
    pose = np.zeros((n - 1, 6))
    for name in ("d_opt", "e_opt", "condition_number", "max_eigen_ratio"):
        for sub in ("all", "trans", "rot"):
            y = dg.apply_degen_function(info, pose.T[:, None, :], sub, name)
            ms, ps = dor.subset(mats, pose, sub)
            np.testing.assert_allclose(y, dor.evaluate(name, ms, ps), rtol=1e-7, atol=1e-9, err_msg=f"{name}/{sub}")
    eng.close()


def test_partitioned_solve_is_deterministic(oracle):
    """The sweep and spike waves of a chunk, and the two teams of the separator chain, hand data to each
    other through LDS counters and barriers: 150 repeated solves of the same systems must give the same bits
    (a stale hand-shake cell once showed up as a last-bit difference in about one run out of 25)."""
    ranges = [(0, N), (5, 150), (0, 64), (10, 47), (0, 133), (2, 199)]
    eng, _ = make_engine(oracle, 0, ranges)
    eng.linearize(0)
    eng.assemble()
    eng.solve()
    first = [eng.read_delta(w, lo, hi - lo).copy() for w, (lo, hi) in enumerate(ranges)]
    assert all(np.isfinite(d).all() for d in first)
    for rep in range(150):
        eng.solve()
        if rep % 10 == 9:
            for w, (lo, hi) in enumerate(ranges):
                np.testing.assert_array_equal(eng.read_delta(w, lo, hi - lo), first[w], err_msg=f"solve {rep} window {w}")
    eng.close()


def test_partitioned_random_window_lengths_and_chunk_counts(oracle):
    """Random window lengths, offsets and chunk counts (chunk boundaries land everywhere relative to the
    between-factor pattern; short windows fall back to fewer chunks): the increment solves the engine's own
    band system to backward-stable accuracy every time."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    rng = np.random.default_rng(2024)
    seq = synth.make_sequence(seed=77, n_kf=N)
    prob = helpers.build_problem(oracle, seq, perturb=0.01)
    worst = 0.0
    for trial in range(12):
        chunks = int(rng.integers(2, 41))
        ranges = []
        for w in range(6):
            n = int(rng.integers(1, N + 1))
            lo = int(rng.integers(0, N - n + 1))
            ranges.append((lo, lo + n))
        eng = Engine(EngineOpts(windows=len(ranges), capacity=N + 8, chunks=chunks))
        for w, (lo, hi) in enumerate(ranges):
            helpers.load_engine(eng, w, prob, lo=lo, hi=hi)
        eng.linearize(0)
        eng.assemble()
        eng.solve()
        for w, (lo, hi) in enumerate(ranges):
            H, g = eng.read_normal(w, lo, hi - lo)
            d = eng.read_delta(w, lo, hi - lo)
            Hl, gl = H.astype(np.longdouble), g.astype(np.longdouble)
            bg = float(np.abs(band_matvec(Hl, np.longdouble(1e-5), d.astype(np.longdouble)) + gl).max() / np.abs(gl).max())
            worst = max(worst, bg)
            assert bg < 1e-9, (chunks, lo, hi, bg)
            assert eng.read_lm(w)["solve_failures"] == 0
        eng.close()
    print("worst backward error over the random cases", worst)
