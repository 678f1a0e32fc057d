"""-m gpu: K1 + K3 fused (k_linearize_assemble), an opt-in form of vf_engine_iterate for whole-window-sweep engines:
the IMU Jacobians stay in LDS, H and g are double-buffered like the states.  It must give what the unfused kernels give
(K1 -> J stream in HBM -> K3): states, costs, H and g -- cold, warm (after slides), with and without marginalisation, on
ragged windows -- and the oracle's trajectory.  The two forms run the same source (linearize_imu_core, assemble_tile),
but the compiler contracts multiply-adds differently in the two instantiations, so they agree to rounding (1e-12), not
bit for bit; fused warm start against fused cold start IS bit for bit.  VF_FUSED=1 / 0 selects the form for small
test batches (read by vf_engine_create)."""
import contextlib
import os

import numpy as np
import pytest

from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def fused(on):
    old = os.environ.get("VF_FUSED")
    os.environ["VF_FUSED"] = str(int(on))          # 0: K1 -> K3; 1: K1 into LDS + MFMA tiles; 2: lane per factor (k_lin_asm_v)
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("VF_FUSED", None)
        else:
            os.environ["VF_FUSED"] = old


def _engine(n_total, ranges, on, seed0=40):
    with fused(on):
        eng = Engine(EngineOpts(windows=len(ranges), capacity=n_total, chunks=1))
    for w, (lo, hi) in enumerate(ranges):
        seq = synth.make_sequence(seed=seed0 + w, n_kf=n_total)
        eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(w, seq.btw_a, seq.btw_b, synth.between_records(seq))
        eng.set_states(w, 0, seq.gt_states[:1])
        eng.set_prior(w, lo, synth.prior_record(seq.gt_states[lo], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
        eng.predict(w, 1, hi - 1)
        # off the optimum, so that the first trials are real steps
        rng = np.random.default_rng(seed0 + w)
        st = eng.get_states(w, 0, hi)
        st[1:, 4:10] += rng.normal(size=(hi - 1, 6)) * 0.02
        eng.set_states(w, 0, st)
        eng.set_range(w, lo, hi)
    return eng


def _same(a, b, ranges, what, exact=False):
    for w, (lo, hi) in enumerate(ranges):
        sa, sb = a.get_states(w, lo, hi - lo), b.get_states(w, lo, hi - lo)
        la, lb = a.read_lm(w), b.read_lm(w)
        if exact:
            assert np.array_equal(sa, sb), (what, w, np.abs(sa - sb).max())
            assert la == lb, (what, w, la, lb)
        else:
            # (a trial at the rounding floor of the cost may be accepted by one form and rejected by the other: the
            # states then differ by that last, sub-nanometre step)
            assert np.abs(sa - sb).max() <= 1e-9, (what, w, np.abs(sa - sb).max())
            assert abs(la["cost"] - lb["cost"]) <= 1e-9 * lb["cost"], (what, w, la, lb)


def _same_normal(a, b, ranges, what, refresh_b=True):
    """a: fused engine (buffer sel of H, g always belongs to the current states).  b: unfused engine, whose single H, g
    are those of the current states only once K3 has run again after an accepted trial: refresh_b runs K1 + K3 on it
    (which also makes its next solve a cold start -- bit-identical to a warm one, tests/test_gpu_warm_start.py)."""
    if refresh_b:
        for w, (lo, hi) in enumerate(ranges):          # same linearisation point for both
            b.set_states(w, lo, a.get_states(w, lo, hi - lo))
        b.linearize(0)
        b.assemble()
    for w, (lo, hi) in enumerate(ranges):
        Ha, ga = a.read_normal(w, lo, hi - lo)
        Hb, gb = b.read_normal(w, lo, hi - lo)
        if not refresh_b:
            assert np.array_equal(ga, gb), (what, w)
        for k in range(hi - lo):
            for d in range(min(k, 3) + 1):      # blocks reaching in front of the window are never read
                if refresh_b:
                    assert np.abs(Ha[k, d] - Hb[k, d]).max() <= 1e-12 * max(np.abs(Hb[k, d]).max(), 1e-300), (what, w, k, d)
                else:
                    assert np.array_equal(Ha[k, d], Hb[k, d]), (what, w, k, d)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("marginalize", [True, False])
def test_fused_equals_unfused(marginalize, mode):
    """Cold solve, then fixed-lag updates (slide + warm solve), on ragged windows whose ends fall on every position of
    the 8-keyframe tiles: bit-identical states, LM bookkeeping, H and g."""
    n_total = 128
    ranges = [(0, 90), (3, 77), (8, 96), (13, 64), (0, 21)]
    f, u = _engine(n_total, ranges, mode), _engine(n_total, ranges, False)
    for e in (f, u):
        e.iterate(5)
    _same(f, u, ranges, "cold")
    _same_normal(f, u, ranges, "cold")
    for s in range(11):
        for e in (f, u):
            e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=marginalize)
            e.iterate(1 + s % 4)
        ranges = [(lo + 1, hi + 1) for lo, hi in ranges]
        _same(f, u, ranges, f"slide {s}")
    _same_normal(f, u, ranges, "after slides")
    # two slides before one solve
    for e in (f, u):
        e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=marginalize)
        e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=marginalize)
        e.iterate(3)
    ranges = [(lo + 2, hi + 2) for lo, hi in ranges]
    _same(f, u, ranges, "double slide")
    f.close(); u.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_fused_warm_start_equals_cold_start(mode):
    n_total = 100
    ranges = [(0, 70), (5, 61)]
    warm, cold = _engine(n_total, ranges, mode), _engine(n_total, ranges, mode)
    for e in (warm, cold):
        e.iterate(4)
    for s in range(9):
        for e in (warm, cold):
            e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
        cold.linearize(0)                 # any stage call makes the next solve a cold start
        cold.assemble()
        warm.iterate(1 + s % 3)
        cold.iterate(1 + s % 3)
        ranges = [(lo + 1, hi + 1) for lo, hi in ranges]
        _same(warm, cold, ranges, f"slide {s}", exact=True)
    _same_normal(warm, cold, ranges, "end", refresh_b=False)
    warm.close(); cold.close()


def test_fused_convergence_exit_and_stage_calls():
    """GTSAM's termination rule on a fused engine (windows drop out of the remaining trials), and the unfused stage
    calls on a fused engine (linearize / assemble / solve / retract / decide keep their meaning and agree with iterate)."""
    n_total = 80
    ranges = [(0, 72), (2, 50), (0, 33)]
    f, u = _engine(n_total, ranges, True), _engine(n_total, ranges, False)
    for e in (f, u):
        e.set_convergence(1e-5, 1e-5)
        e.iterate(8)
    _same(f, u, ranges, "convergence exit")
    assert f.read_lm(0)["accepted"] + f.read_lm(0)["rejected"] < 8
    f.close(); u.close()
    f, u = _engine(n_total, ranges, True), _engine(n_total, ranges, True)
    u.iterate(3)
    f.linearize(0); f.decide(init=True)
    for _ in range(3):
        f.assemble(); f.solve(); f.retract(); f.linearize(1); f.decide()
    for w, (lo, hi) in enumerate(ranges):
        assert np.abs(f.get_states(w, lo, hi - lo) - u.get_states(w, lo, hi - lo)).max() <= 1e-9, w
    f.close(); u.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_large_batch_runs_fused_and_matches_the_oracle(oracle, mode):
    """160 windows, fused: LM trajectories of a few of them against the oracle, and the fused stage timer exists only
    on fused engines."""
    from vil_sensor_fusion_amd._lib import VilFusionError
    n, B = 150, 160
    with fused(mode):
        eng = Engine(EngineOpts(windows=B, capacity=n + 10))
    picks = {0: (0, n), 77: (3, 131), 159: (17, n), 80: (0, 40)}
    probs = {}
    for w, (lo, hi) in picks.items():
        seq = synth.make_sequence(seed=300 + w, n_kf=n)
        probs[w] = helpers.build_problem(oracle, seq, perturb=0.01)
        helpers.load_engine(eng, w, probs[w], lo=lo, hi=hi)
    eng.iterate(5)
    for w, (lo, hi) in picks.items():
        win = helpers.oracle_window(oracle, probs[w], lo=lo, hi=hi)
        costs, acc, _ = win.lm(iterations=5)
        ate, rot = helpers.ate(eng.get_states(w, lo, hi - lo), win.states)
        lm = eng.read_lm(w)
        print(f"window {w}: ATE {ate:.3e} m, rot {rot:.3e} rad, cost {lm['cost']:.6e} vs {costs[-1]:.6e}")
        assert ate <= 1e-6 and rot <= 1e-6
        # (accept / reject of a trial at the rounding floor of the cost is decided by its last bit: counts may differ by one)
        assert abs(lm["cost"] - costs[-1]) <= 1e-9 * costs[-1] and abs(lm["accepted"] - int(np.sum(np.array(acc) == 1))) <= 1
    assert eng.time_stage("linearize_assemble", reps=1) > 0
    eng.close()
    small = Engine(EngineOpts(windows=2, capacity=64))
    with pytest.raises(VilFusionError):
        small.time_stage("linearize_assemble", reps=1)
    small.close()
