"""-m gpu: hybrid K4 under the LM termination rule.  A batch of whole-window sweeps launches both forms of the solve in
every trial; on the device, the sweep runs while more than `gate_T` windows still take trials and the partitioned form
(K4p, chunks joined by 27-dof separators) once fewer are left (the sweep's launch time is one window's dependency chain
whatever their number).  Same factorisation in another elimination order: trajectories agree to rounding with the
sweep-only engine and with the oracle under the same rule."""
import numpy as np
import pytest

from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth

pytestmark = pytest.mark.gpu


def _run(threshold, probs, picks, n, B, iters, active_list=None):
    """threshold = vf_engine_opts.hybrid_threshold (None: never the partitioned form)"""
    eng = Engine(EngineOpts(windows=B, capacity=n + 10, hybrid_threshold=-1 if threshold is None else threshold, hybrid_active_list=active_list))
    for w, (lo, hi) in picks.items():
        helpers.load_engine(eng, w, probs[w], lo=lo, hi=hi)
    eng.set_convergence(1e-5, 1e-5)
    eng.iterate(iters)
    out = {w: (eng.get_states(w, lo, hi - lo), eng.read_lm(w)) for w, (lo, hi) in picks.items()}
    eng.close()
    return out


def test_hybrid_solve_matches_sweeps_and_oracle(oracle):
    n, B, iters = 150, 160, 8
    picks = {0: (0, n), 77: (3, 131), 159: (17, n), 80: (0, 40), 5: (0, 96)}
    probs = {}
    for w in picks:
        seq = synth.make_sequence(seed=500 + w, n_kf=n)
        probs[w] = helpers.build_problem(oracle, seq, perturb=0.003 if w % 2 else 0.03)
    sweep = _run(None, probs, picks, n, B, iters)          # sweeps only
    always = _run(100000, probs, picks, n, B, iters)       # the partitioned form from the first trial on
    mixed = _run(3, probs, picks, n, B, iters)             # sweeps until <= 3 of the 5 windows are left
    for w, (lo, hi) in picks.items():
        win = helpers.oracle_window(oracle, probs[w], lo=lo, hi=hi)
        costs, acc, _ = win.lm(iterations=iters, rel_tol=1e-5, abs_tol=1e-5)
        trials = int(np.sum(np.array(acc) >= 0))
        for name, res in (("sweep", sweep), ("partitioned", always), ("mixed", mixed)):
            st, lm = res[w]
            ate, rot = helpers.ate(st, win.states)
            assert ate <= 1e-6 and rot <= 1e-6, (name, w, ate, rot)
            assert lm["solve_failures"] == 0 and abs(lm["cost"] - costs[-1]) <= 1e-9 * costs[-1], (name, w)
            assert abs(lm["accepted"] + lm["rejected"] - trials) <= 1, (name, w, lm, trials)
        assert np.abs(mixed[w][0] - sweep[w][0]).max() <= 1e-9 and np.abs(always[w][0] - sweep[w][0]).max() <= 1e-9
    assert any(sweep[w][1]["accepted"] + sweep[w][1]["rejected"] < iters for w in picks)      # the rule did stop some windows early


def test_compacted_active_list_changes_nothing_but_the_dispatch(oracle):
    """vf_engine_opts.hybrid_active_list: under the termination rule the hybrid's one-wave sweeps take their windows from the
    compacted list of those still taking trials (k_count_active), so that the active ones are dispatched first; window i is
    then no longer workgroup i -- same windows, same arithmetic, the same bits as with the list off."""
    n, B, iters = 120, 200, 8
    picks = {w: (0, n - (w % 7)) for w in (0, 3, 50, 51, 52, 120, 199)}
    probs = {w: helpers.build_problem(oracle, synth.make_sequence(seed=700 + w, n_kf=n), perturb=0.002 * (1 + w % 5)) for w in picks}
    on = _run(3, probs, picks, n, B, iters, active_list=1)
    off = _run(3, probs, picks, n, B, iters, active_list=0)
    for w in picks:
        np.testing.assert_array_equal(on[w][0], off[w][0])
        assert on[w][1] == off[w][1]
    assert len({on[w][1]["accepted"] + on[w][1]["rejected"] for w in picks}) > 1         # windows did stop at different trials
