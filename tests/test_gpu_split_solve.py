"""-m gpu: the split form of the one-wave band solve (vf_engine_opts.solve_split_min: k_band_forward with the compact 20 KB
trailing window at two waves per SIMD, then k_band_backward) does the arithmetic of the fused k_band_solve in the same
order: states, LM counters, panels and marginal priors must be identical bit for bit -- over ragged windows, a
marginalised slide (6 x 15 strip of the marginal prior on the third keyframe), identity-padded window lengths and the
LM termination rule (windows that drop out of later trials)."""
import numpy as np
import pytest

from tests.test_gpu_ingest import _engine, _feed
from vil_sensor_fusion_amd import synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS

pytestmark = pytest.mark.gpu


def test_split_solve_is_bit_identical_to_the_fused_kernel():
    n, updates, B = 150, 4, 9
    seqs = [synth.make_sequence(seed=700 + i, n_kf=n + updates + 1) for i in range(B)]
    common = dict(chunks=1, sweep_two_sided_max=0)
    fused = _engine(None, seqs, n, updates, solve_split_min=0, **common)
    split = _engine(None, seqs, n, updates, solve_split_min=1, **common)
    for e in (fused, split):
        for w in range(B):
            e.set_range(w, 0, n - 8 - 7 * (w % 4) - (w % 3))    # ragged: lengths not multiples of 4 -> identity padding rows; the
            #                                                  keyframes the two slides append have their factors resident
        e.iterate(15)

    def same(u0):
        for w in range(B):
            m = n - 8 - 7 * (w % 4) - (w % 3)
            np.testing.assert_array_equal(fused.get_states(w, u0, m), split.get_states(w, u0, m))
            assert fused.read_lm(w) == split.read_lm(w)
            np.testing.assert_array_equal(fused.read_panels(w, u0, m), split.read_panels(w, u0, m))
            np.testing.assert_array_equal(fused.read_delta(w, u0, m), split.read_delta(w, u0, m))

    same(0)
    for e in (fused, split):
        e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)      # (the appended keyframe's factors are resident up to n)
        e.iterate(5)
    same(1)
    a, b = fused.read_marginal(3), split.read_marginal(3)
    np.testing.assert_array_equal(a["L"], b["L"])
    for e in (fused, split):
        e.set_convergence(1e-5, 1e-5)
        e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
        e.iterate(5)
    same(2)
    fused.close()
    split.close()


def test_default_threshold_uses_the_split_form_from_2048_windows_on():
    """vf_engine_opts.solve_split_min defaults to 2048: a 2048-window engine left at that default (and with the assembling
    sweep, which would take precedence, switched off) runs k_band_forward + k_band_backward, also behind the termination
    rule's gate (hybrid K4), and gives the bits of the fused kernel."""
    from vil_sensor_fusion_amd import Engine, EngineOpts
    n, B = 48, 2048
    seqs = [synth.make_sequence(seed=720 + i, n_kf=n + 2) for i in range(4)]
    engines = []
    for split in (None, 0):                       # library default (2048) / never
        eng = Engine(EngineOpts(windows=B, capacity=n + 2, solve_split_min=split, solve_assemble_min=0))
        recs = [synth.between_records(s) for s in seqs]
        for w in range(B):
            s = seqs[w % 4]
            eng.preintegrate(w, 1, s.imu_off[1:n + 1], s.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
            m = s.btw_b < n
            eng.set_between(w, s.btw_a[m], s.btw_b[m], recs[w % 4][m])
            eng.set_states(w, 0, s.gt_states[0].reshape(1, 16))
            eng.set_prior(w, 0, synth.prior_record(s.gt_states[0], REFERENCE_PRIOR_SIGMAS))
            eng.set_range(w, 0, 1)
        eng.predict(-1, 1, n - 1)
        for w in range(B):
            eng.set_range(w, 0, n - (w % 5))
        eng.iterate(6)
        eng.set_convergence(1e-5, 1e-5)
        eng.iterate(4)
        engines.append(eng)
    a, b = engines
    for w in (0, 1, 517, 1023, 2047):
        m = n - (w % 5)
        np.testing.assert_array_equal(a.get_states(w, 0, m), b.get_states(w, 0, m))
        assert a.read_lm(w) == b.read_lm(w)
    assert a.time_stage("solve", 2) > 0
    a.close()
    b.close()
