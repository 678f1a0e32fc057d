"""Warm start of the fixed-lag update (k_linearize_tail + ends-only assembly): after a solve followed by nothing but
slides, vf_engine_iterate linearises only the appended keyframes' factors and re-assembles only the rows near the two
ends of windows whose last trial was rejected.  It must be indistinguishable from the cold path (everything linearised
and assembled again): states, costs, H and g bit for bit."""
import numpy as np
import pytest

from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS

pytestmark = pytest.mark.gpu


def _engine(n_total, n0, windows, chunks):
    eng = Engine(EngineOpts(windows=windows, capacity=n_total, chunks=chunks))
    for w in range(windows):
        seq = synth.make_sequence(seed=11 + w, n_kf=n_total)
        eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(w, seq.btw_a, seq.btw_b, synth.between_records(seq))
        eng.set_states(w, 0, seq.gt_states[:1])
        eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
        eng.predict(w, 1, n0 - 1)
        eng.set_range(w, 0, n0)
    return eng


@pytest.mark.parametrize("chunks", [1, 0])          # one sweep per window / partitioned solve
@pytest.mark.parametrize("marginalize", [True, False])
def test_warm_start_equals_cold_start(chunks, marginalize):
    n0, slides, windows = 70, 12, 3
    warm = _engine(n0 + slides + 2, n0, windows, chunks)
    cold = _engine(n0 + slides + 2, n0, windows, chunks)
    for e in (warm, cold):
        e.iterate(4)
    for s in range(slides):
        warm.slide(REFERENCE_PRIOR_SIGMAS, marginalize=marginalize)
        cold.slide(REFERENCE_PRIOR_SIGMAS, marginalize=marginalize)
        cold.linearize(0)                 # any stage call makes the next solve a cold start
        k = 1 + s % 4                     # odd and even trial counts: last trials accepted and rejected both occur
        warm.iterate(k)
        cold.iterate(k)
        for w in range(windows):
            lo, hi = s + 1, n0 + s + 1
            a, b = warm.get_states(w, lo, hi - lo), cold.get_states(w, lo, hi - lo)
            assert np.array_equal(a, b), (s, w, np.abs(a - b).max())
            la, lb = warm.read_lm(w), cold.read_lm(w)
            assert la["cost"] == lb["cost"] and la["accepted"] == lb["accepted"] and la["rejected"] == lb["rejected"], (s, w, la, lb)
    # H and g as the last solve saw them (re-assembled where needed only / everywhere)
    for w in range(windows):
        lo, hi = slides, n0 + slides
        Ha, ga = warm.read_normal(w, lo, hi - lo)
        Hb, gb = cold.read_normal(w, lo, hi - lo)
        assert np.array_equal(ga, gb)
        # blocks reaching in front of the window are never read by the solver and may hold older values
        for k in range(hi - lo):
            for d in range(4):
                if k - d >= 0:
                    assert np.array_equal(Ha[k, d], Hb[k, d]), (w, k, d)


def test_two_slides_before_a_solve():
    n0 = 64
    warm = _engine(n0 + 12, n0, 2, 1)
    cold = _engine(n0 + 12, n0, 2, 1)
    for e in (warm, cold):
        e.iterate(3)
    for s in range(4):
        for e in (warm, cold):
            e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
            e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
        cold.linearize(0)
        warm.iterate(3)
        cold.iterate(3)
        for w in range(2):
            lo, hi = 2 * (s + 1), n0 + 2 * (s + 1)
            assert np.array_equal(warm.get_states(w, lo, hi - lo), cold.get_states(w, lo, hi - lo)), (s, w)
