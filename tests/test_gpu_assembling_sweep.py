"""-m gpu: the assembling forward sweep (vf_engine_opts.solve_assemble_min; k_band_forward_asm + k_band_backward).

From that many windows on, the one-wave band solver forms the block rows of the normal equations itself -- from the J
stream K1 leaves, the between linearisations of K2, the prior and the marginal prior -- on the matrix cores, and K3
(k_assemble) is not launched: H is neither written nor read back.  The sums are the same, their order is not, so this
form agrees with the two-kernel one to rounding (1e-12 relative on the Cholesky panels), not to the bit; against the
oracle it is held to the same bars as every other form.  Covered here: ragged windows whose first slot moves through
all eight positions of a J-stream tile (slides), marginalised slides (6 x 15 strip of the marginal prior), a prior on
a keyframe in the middle of the window, windows without any between factor and with factors reaching back 1, 2 and 3
keyframes, vf_engine_read_normal on such an engine, and the cases in which the engine must NOT use the form (far
factors; the partitioned half of the termination rule's hybrid solve), where it has to give the bits of the two-kernel path."""
import numpy as np
import pytest

from tests import helpers
from tests.test_gpu_ingest import _engine, _feed
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS

pytestmark = pytest.mark.gpu
SWEEP = dict(chunks=1, sweep_two_sided_max=0)          # one wave per window whatever the batch
WAVES = pytest.mark.parametrize("waves", [1, 2])       # the sweep as one wave per window / as eliminator + assembler wave (solve_assemble_waves)


def _pair(seqs, n, updates, waves=1, **opts):
    return (_engine(None, seqs, n, updates, solve_assemble_min=0, **SWEEP, **opts),
            _engine(None, seqs, n, updates, solve_assemble_min=1, solve_assemble_waves=waves, **SWEEP, **opts))


@WAVES
def test_first_trial_agrees_with_the_two_kernel_form_to_rounding(waves):
    """One staged LM trial from the same linearisation: increments and Cholesky panels of K3 + k_band_solve against the
    assembling sweep, ragged windows (identity-padded lengths), first slots 0 .. 8 (every position inside a tile), the prior on the
    first keyframe or on the twelfth."""
    n, B = 150, 10
    seqs = [synth.make_sequence(seed=810 + i, n_kf=n + 2) for i in range(B)]
    two, asm = _pair(seqs, n, 0, waves)
    worst_p = worst_d = 0.0
    for e in (two, asm):
        for w in range(B):
            pk = w % 9 + (11 if w % 2 else 0)        # odd windows: the prior sits on a keyframe in the middle (rows past the first four take it by another path)
            e.set_prior(w, pk, synth.prior_record(seqs[w].gt_states[pk], REFERENCE_PRIOR_SIGMAS))
            e.set_range(w, w % 9, n - 5 * (w % 4) - (w % 3))
        e.linearize()
        e.decide(init=True)
        e.assemble()
        e.solve()
    for w in range(B):
        lo, hi = w % 9, n - 5 * (w % 4) - (w % 3)
        pa, pb = two.read_panels(w, lo, hi - lo), asm.read_panels(w, lo, hi - lo)
        da, db = two.read_delta(w, lo, hi - lo), asm.read_delta(w, lo, hi - lo)
        worst_p = max(worst_p, np.abs(pa - pb).max() / np.abs(pa).max())
        worst_d = max(worst_d, np.abs(da - db).max() / np.abs(da).max())
        assert asm.read_lm(w)["solve_failures"] == 0
    print(f"assembling sweep vs K3 + k_band_solve, first trial: panels {worst_p:.2e}, increments {worst_d:.2e} (relative to the largest entry)")
    # (the increments carry the condition number of the window, 1e11: two roundings of H differ by 1e-16 * 1e11 there)
    assert worst_p <= 1e-10 and worst_d <= 1e-4
    two.close()
    asm.close()


@WAVES
def test_fixed_lag_updates_with_ingest_match_the_oracle(oracle, waves):
    """bench.py's update -- ingest (K0 at the current bias) + marginalised slide + 5 LM trials -- for 12 updates, so that the
    window's first slot passes through every position of a J-stream tile and the tile switch falls on every phase of the
    sweep; windows of different lengths; against helpers.FixedLagOracle doing the same."""
    n, updates, B = 160, 12, 5
    seqs = [synth.make_sequence(seed=830 + i, n_kf=n + updates + 2) for i in range(B)]
    eng = _engine(None, seqs, n, updates, solve_assemble_min=1, solve_assemble_waves=waves, **SWEEP)
    eng.iterate(60)
    prm = oracle.carla_imu_params()
    refs = []
    for w, seq in enumerate(seqs):
        prob = helpers.build_problem(oracle, seq)
        refs.append(helpers.FixedLagOracle(oracle, prob, n, 5, init_iterations=60, ingest=(seq, prm)))
        a, r = helpers.ate(eng.get_states(w, 0, n), refs[w].window_states)
        assert a <= 1e-8 and r <= 1e-6, ("initial solve", w, a, r)
    worst = 0.0
    for u in range(1, updates + 1):
        eng.ingest_tail(*_feed(seqs, n + u - 1))
        eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
        eng.iterate(5)
        for w in range(B):
            st = refs[w].update()
            a, r = helpers.ate(eng.get_states(w, u, n), st)
            worst = max(worst, a)
            assert a <= 1e-8 and r <= 1e-6, (u, w, a, r)
            lm = eng.read_lm(w)
            assert lm["solve_failures"] == 0
            assert abs(lm["cost"] - refs[w].costs[-1]) <= 1e-9 * abs(refs[w].costs[-1])
    eng.ingest_status()
    got, exp = eng.read_marginal(2), refs[2].marg.arrays()
    np.testing.assert_allclose(got["L"], exp["L"], atol=1e-9 * np.abs(exp["L"]).max())
    print(f"assembling sweep, {updates} fixed-lag updates with ingest, {B} windows: worst ATE vs the oracle {worst:.3e} m")
    eng.close()


@WAVES
def test_between_factor_spans_and_gaps(oracle, waves):
    """Between factors reaching back one, two and three keyframes, keyframes without any, and an IMU-only stretch: the older
    keyframe's term goes into a row that is already in the trailing window, at a distance the sweep only learns from the
    factor -- against the oracle on the same graph."""
    n = 90
    seq = synth.make_sequence(seed=851, n_kf=n)
    prob = helpers.build_problem(oracle, seq)
    keep = np.ones(prob["btw_a"].size, dtype=bool)
    keep[(prob["btw_b"] >= 30) & (prob["btw_b"] < 41)] = False          # IMU only
    keep[prob["btw_b"] % 7 == 3] = False                                 # single keyframes without a factor
    spans = prob["btw_b"][keep] - prob["btw_a"][keep]
    assert set(np.unique(spans)) >= {1, 2, 3}
    p2 = dict(prob, btw_a=prob["btw_a"][keep], btw_b=prob["btw_b"][keep], btw=prob["btw"][keep])
    eng = Engine(EngineOpts(windows=3, capacity=n, solve_assemble_min=1, solve_assemble_waves=waves, **SWEEP))
    for w in range(3):
        helpers.load_engine(eng, w, p2, lo=0, hi=n - w)
    eng.iterate(25)
    for w in range(3):
        win = helpers.oracle_window(oracle, p2, 0, n - w)
        win.lm(iterations=25)
        a, r = helpers.ate(eng.get_states(w, 0, n - w), win.states)
        assert a <= 1e-9 and r <= 1e-6, (w, a, r)
    eng.close()


@WAVES
@pytest.mark.parametrize("n", [2, 3, 4, 5, 7, 9, 13])
def test_tiny_windows(oracle, n, waves):
    """windows shorter than the sweep's unroll, than a J-stream tile, than the profile: rows past the window's end are
    identity rows, factors past it zeros."""
    seq = synth.make_sequence(21, 24)
    prob = helpers.build_problem(oracle, seq, perturb=0.01)
    eng = Engine(EngineOpts(windows=2, capacity=24, solve_assemble_min=1, solve_assemble_waves=waves, **SWEEP))
    for w, lo in ((0, 0), (1, 6)):
        helpers.load_engine(eng, w, prob, lo=lo, hi=lo + n)
    assert eng.solve_form() == "assembling"
    eng.iterate(12)
    for w, lo in ((0, 0), (1, 6)):
        win = helpers.oracle_window(oracle, prob, lo=lo, hi=lo + n)
        costs, _, _ = win.lm(iterations=12)
        a, r = helpers.ate(eng.get_states(w, lo, n), win.states)
        assert a <= 1e-6 and r <= 1e-6, (n, w, a, r)
        assert abs(eng.read_lm(w)["cost"] - costs[-1]) <= 1e-6 * max(costs[-1], 1e-9)
    eng.close()


@WAVES
def test_termination_rule_on_a_small_batch(waves):
    """GTSAM's LM rule on an engine too small for the hybrid form (<= 128 windows): windows that are done drop out of the later
    trials of the assembling sweep as of every other kernel; same trial counts as the two-kernel form, states to rounding."""
    n, B = 110, 6
    seqs = [synth.make_sequence(seed=890 + i, n_kf=n + 2) for i in range(B)]
    two, asm = _pair(seqs, n, 0, waves)
    for e in (two, asm):
        e.set_convergence(1e-5, 1e-5)
        e.iterate(30)
    assert asm.solve_form() == "assembling" and two.solve_form() == "one_wave"
    for w in range(B):
        a, b = two.read_lm(w), asm.read_lm(w)
        assert a["accepted"] + a["rejected"] == b["accepted"] + b["rejected"] < 30
        assert np.abs(two.get_states(w, 0, n) - asm.get_states(w, 0, n)).max() <= 1e-9
    two.close()
    asm.close()


def test_read_normal_assembles_on_demand():
    """vf_engine_read_normal on an engine whose solves never store H: the rows come from K3, launched for the read, and are
    the rows the two-kernel engine holds after the same trials."""
    n, B = 70, 4
    seqs = [synth.make_sequence(seed=860 + i, n_kf=n + 2) for i in range(B)]
    two, asm = _pair(seqs, n, 0)
    for e in (two, asm):
        e.linearize()
        e.decide(init=True)
        e.assemble()                     # (nothing to do on the assembling engine)
    for w in range(B):
        Ha, ga = two.read_normal(w, 0, n)
        Hb, gb = asm.read_normal(w, 0, n)
        assert np.abs(Hb).max() > 0
        np.testing.assert_array_equal(Ha, Hb)          # same states, same linearisation, the same K3
        np.testing.assert_array_equal(ga, gb)
    for e in (two, asm):
        e.iterate(8)                     # (trials are rejected on the way: the read must not depend on the `fresh` flags they leave)
    for w in range(B):
        Ha, _ = two.read_normal(w, 0, n)
        Hb, _ = asm.read_normal(w, 0, n)
        # the states of the two engines differ by rounding by now (1e-14 m), hence their linearisations
        assert np.abs(Ha - Hb).max() <= 1e-6 * np.abs(Ha).max()
    two.close()
    asm.close()


def test_read_normal_of_windows_the_termination_rule_has_finished():
    """... and with the termination rule on (vf_engine_set_convergence): after a solve every converged window has done = 1, and
    K3 skips such windows during a solve.  A read of H must assemble them all the same -- Engine.pose_information, the
    per-keyframe 6x6 blocks the degeneracy metrics run on (BASELINE configs[2]), is asked for exactly those.  The rows must be
    those of the two-kernel engine (which wrote H on its last accepted trial) to the rounding of their states."""
    n, B = 70, 4
    seqs = [synth.make_sequence(seed=870 + i, n_kf=n + 2) for i in range(B)]
    two, asm = _pair(seqs, n, 0)
    for e in (two, asm):
        e.set_convergence(1e-5, 1e-5)
        e.iterate(40)
    assert asm.solve_form() in ("assembling", "hybrid")
    for w in range(B):
        Ha, ga = two.read_normal(w, 0, n)
        Hb, gb = asm.read_normal(w, 0, n)
        assert np.abs(Hb).max() > 0 and np.abs(Hb[:, 0, :6, :6]).min(axis=(1, 2)).max() >= 0
        assert np.abs(Ha - Hb).max() <= 1e-6 * np.abs(Ha).max(), w
        Pa, Pb = two.pose_information(w, 0, n), asm.pose_information(w, 0, n)
        assert np.abs(Pb).max() > 0 and np.abs(Pa - Pb).max() <= 1e-6 * np.abs(Pa).max()
        # the read is repeatable and leaves the solve state alone
        Hc, _ = asm.read_normal(w, 0, n)
        np.testing.assert_array_equal(Hb, Hc)
    lm0 = asm.read_lm(0)
    asm.iterate(2)
    assert asm.read_lm(0)["solve_failures"] == lm0["solve_failures"] == 0
    two.close()
    asm.close()


def test_forms_that_need_H_fall_back_to_the_two_kernel_path():
    """Far between factors (their low-rank correction solves from H and g): an engine asked for the assembling sweep must run
    K3 + K4 there and give their bits.  The termination rule's hybrid solve: its partitioned half reads H, which K3 then
    assembles for it alone (gated, and without trusting the `fresh` flags) -- below the hybrid threshold that half does
    all the work, and the bits are those of the two-kernel engine.  (The sweep half under the rule: test_gpu_headline_path.)"""
    n, B = 80, 3
    seqs = [synth.make_sequence(seed=870 + i, n_kf=n + 4) for i in range(B)]
    two, asm = _pair(seqs, n, 2)
    rec = synth.between_records(seqs[0])
    for e in (two, asm):
        e.set_extra_between(1, np.array([5], dtype=np.int32), np.array([60], dtype=np.int32), rec[:1])
        e.iterate(10)
    for w in range(B):
        np.testing.assert_array_equal(two.get_states(w, 0, n), asm.get_states(w, 0, n))
    for e in (two, asm):
        e.set_extra_between(1, np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros((0, 28)))
    two.close()
    asm.close()
    # the hybrid form exists on batches above 128 windows only
    n, B = 40, 130
    seqs = [synth.make_sequence(seed=880 + i, n_kf=n + 2) for i in range(3)]
    engines = []
    for a in (0, 1):
        eng = Engine(EngineOpts(windows=B, capacity=n + 2, solve_assemble_min=a, sweep_two_sided_max=0))
        recs = [synth.between_records(s) for s in seqs]
        for w in range(B):
            s = seqs[w % 3]
            eng.preintegrate(w, 1, s.imu_off[1:n + 1], s.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
            m = s.btw_b < n
            eng.set_between(w, s.btw_a[m], s.btw_b[m], recs[w % 3][m])
            eng.set_states(w, 0, s.gt_states[0].reshape(1, 16))
            eng.set_prior(w, 0, synth.prior_record(s.gt_states[0], REFERENCE_PRIOR_SIGMAS))
            eng.set_range(w, 0, 1)
        eng.predict(-1, 1, n - 1)
        for w in range(B):
            eng.set_range(w, 0, n)
        eng.set_convergence(1e-5, 1e-5)
        eng.iterate(8)
        engines.append(eng)
    for w in (0, 64, 129):
        np.testing.assert_array_equal(engines[0].get_states(w, 0, n), engines[1].get_states(w, 0, n))
    for e in engines:
        e.close()


def test_library_defaults_pick_the_form_by_batch():
    """vf_engine_solve_form on engines left at the library's defaults: the partitioned form up to 128 windows, two waves per
    window up to 256, one wave per window above, the assembling sweep from 768 on (bench.py's headline batch is 1 024), the hybrid once the termination rule is
    switched on (above 128 windows); chunks >= 2 is the partitioned form whatever the batch."""
    for windows, form in ((2, "partitioned"), (128, "partitioned"), (129, "two_sided"), (256, "two_sided"), (300, "one_wave"), (767, "one_wave"),
                          (768, "assembling"), (1024, "assembling"), (2048, "assembling")):
        eng = Engine(EngineOpts(windows=windows, capacity=64))
        assert eng.solve_form() == form, (windows, eng.solve_form())
        if windows in (300, 1024):       # (the hybrid's sweep half is the assembling one from 768 windows on as well)
            eng.set_convergence(1e-5, 1e-5)
            assert eng.solve_form() == "hybrid"
        eng.close()
    eng = Engine(EngineOpts(windows=1024, capacity=64, chunks=4))
    assert eng.solve_form() == "partitioned"
    eng.close()
    eng = Engine(EngineOpts(windows=2048, capacity=64, solve_assemble_min=0))
    assert eng.solve_form() == "one_wave_split"
    eng.close()


def test_two_waves_per_window_give_the_bits_of_one():
    """k_band_forward_asm2 (eliminator wave + assembler wave on one LDS image, hand-shake through three LDS cells) issues the
    instructions of k_band_forward_asm, row by row, in the same order: states, LM counters, panels and marginal priors are
    identical bit for bit -- ragged windows, marginalised slides that move the first slot through a J-stream tile."""
    n, updates, B = 140, 4, 7
    seqs = [synth.make_sequence(seed=900 + i, n_kf=n + updates + 2) for i in range(B)]
    one = _engine(None, seqs, n, updates, solve_assemble_min=1, solve_assemble_waves=1, **SWEEP)
    two = _engine(None, seqs, n, updates, solve_assemble_min=1, solve_assemble_waves=2, **SWEEP)
    for e in (one, two):
        for w in range(B):
            e.set_range(w, 0, n - 8 - (w % 4))        # (the keyframes the slides append have their factors resident)
        e.iterate(25)
    for w in range(B):
        m = n - 8 - (w % 4)
        np.testing.assert_array_equal(one.get_states(w, 0, m), two.get_states(w, 0, m))
        assert one.read_lm(w) == two.read_lm(w)
        np.testing.assert_array_equal(one.read_panels(w, 0, m), two.read_panels(w, 0, m))
    for u in range(1, 4):
        for e in (one, two):
            e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
            e.iterate(5)
        for w in range(B):
            m = n - 8 - (w % 4)
            np.testing.assert_array_equal(one.get_states(w, u, m), two.get_states(w, u, m))
    a, b = one.read_marginal(3), two.read_marginal(3)
    np.testing.assert_array_equal(a["L"], b["L"])
    one.close()
    two.close()


def test_swapped_roles_are_the_same_sweep():
    """The two waves of a workgroup of the two-wave sweep take their roles from a per-CU agreement (one eliminator per SIMD,
    DESIGN.md 7.16): wave 1 eliminates where wave 0's SIMD already has an eliminator.  Where that happens depends on the
    dispatcher, so the swapped path is exercised by a build in which EVERY workgroup swaps (tools/variants/
    libvilfusion_swapped.so, -DVF_ASM2_ROLES=2, made by __graft_entry__.build()): the same states and LM counters as the
    product library, bit for bit, over LM trials and marginalised slides of ragged windows.  Each library in a child process."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variant = os.path.join(root, "tools", "variants", "libvilfusion_swapped.so")
    if not os.path.exists(variant):
        pytest.fail(f"{variant} missing: run __graft_entry__.build()")
    child = r'''
import json, sys, hashlib
sys.path.insert(0, %(root)r)
if sys.argv[1] != "product":
    from vil_sensor_fusion_amd import _lib
    _lib._SO = %(variant)r
import numpy as np
from tests.test_gpu_ingest import _engine
from vil_sensor_fusion_amd import synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
n, B = 130, 12
seqs = [synth.make_sequence(seed=830 + i, n_kf=n + 8) for i in range(B)]
eng = _engine(None, seqs, n, 6, solve_assemble_min=1, solve_assemble_waves=2, chunks=1, sweep_two_sided_max=0)
for w in range(B):
    eng.set_range(w, 0, n - 8 - (w %% 7))
assert eng.solve_form() == "assembling"
eng.iterate(8)
h = hashlib.sha256()
for s in range(1, 5):
    eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
    eng.iterate(4)
for w in range(B):
    h.update(eng.get_states(w, 4, n - 8 - (w %% 7)).tobytes())
print("RESULT " + json.dumps({"sha": h.hexdigest(), "lm": [eng.read_lm(w) for w in range(B)]}))
''' % {"root": root, "variant": variant}
    outs = []
    for which in ("product", "swapped"):
        res = subprocess.run([sys.executable, "-c", child, which], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-3000:]
        outs.append(json.loads([l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1][7:]))
    assert outs[0] == outs[1] and all(lm["solve_failures"] == 0 and lm["accepted"] > 0 for lm in outs[0]["lm"])
