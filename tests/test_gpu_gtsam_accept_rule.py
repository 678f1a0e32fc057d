"""-m gpu: GTSAM's LM accept rule as an option (vf_engine_opts.min_model_fidelity; VERDICT r5 missing #5).  The reference leaves
`LevenbergMarquardtOptimizer` commented out at GraphManager.cpp:128-129; GTSAM accepts a trial on modelFidelity = (actual
decrease) / (decrease the linearised problem predicts) > minModelFidelity (1e-3), the library by default on its own
relative-decrease test.  With the option on, the device and the oracle apply GTSAM's rule: same decisions, same states."""
import numpy as np
import pytest

from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("perturb", [0.01, 0.08])
def test_gtsam_rule_matches_the_oracle_under_the_same_rule(oracle, perturb):
    n, K = 150, 9
    seq = synth.make_sequence(seed=61, n_kf=n)
    prob = helpers.build_problem(oracle, seq, perturb=perturb)
    eng = Engine(EngineOpts(windows=1, capacity=n + 8, min_model_fidelity=1e-3))
    helpers.load_engine(eng, 0, prob)
    eng.iterate(K)
    lm = eng.read_lm(0)
    win = helpers.oracle_window(oracle, prob)
    costs, acc, lam = win.lm(iterations=K, min_model_fidelity=1e-3)
    ate, rot = helpers.ate(eng.get_states(0, 0, n), win.states)
    print(f"perturbation {perturb}: GTSAM's rule, device {lm['accepted']} accepted / {lm['rejected']} rejected, oracle {int((acc == 1).sum())} / {int((acc == 0).sum())}; "
          f"cost {lm['cost']:.9e} vs {costs[-1]:.9e}; ATE {ate:.2e} m; lambda {lm['lam']:.1e} vs {lam:.1e}")
    assert lm["accepted"] == int((acc == 1).sum()) and lm["rejected"] == int((acc == 0).sum())
    assert ate <= 1e-8 and rot <= 1e-7 and abs(lm["cost"] - costs[-1]) <= 1e-9 * max(1.0, costs[-1])
    assert np.isclose(lm["lam"], lam, rtol=1e-12)
    eng.close()


def test_the_rule_changes_decisions_not_the_optimum(oracle):
    """a start far enough out that the two rules disagree on some trial; both end at the same optimum"""
    n = 120
    seq = synth.make_sequence(seed=62, n_kf=n)
    prob = helpers.build_problem(oracle, seq, perturb=0.08)
    out = {}
    for name, kw in (("own", {}), ("gtsam", dict(min_model_fidelity=1e-3))):
        eng = Engine(EngineOpts(windows=1, capacity=n + 8, **kw))
        helpers.load_engine(eng, 0, prob)
        eng.iterate(25)
        out[name] = (eng.get_states(0, 0, n), eng.read_lm(0))
        eng.close()
    ate, rot = helpers.ate(out["own"][0], out["gtsam"][0])
    print(f"own rule: {out['own'][1]['accepted']} accepted / {out['own'][1]['rejected']} rejected; GTSAM's: {out['gtsam'][1]['accepted']} / {out['gtsam'][1]['rejected']}; "
          f"optima {ate:.2e} m apart")
    assert ate <= 1e-7


def test_graph_manager_takes_the_option():
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    seq = synth.make_sequence(seed=63, n_kf=60)
    outs = []
    for kw in ({}, dict(min_model_fidelity=1e-3)):
        gm = GraphManager(capacity=128, iterations=6, rel_tol=0.0, abs_tol=0.0, **kw)
        gm.setInitialState(seq.gt_states[0])
        gm.addIMUMeasurement(0.0, seq.imu_steps[0, 1:4], seq.imu_steps[0, 4:7])
        t = 0.0
        for k in range(1, 60):
            for s in seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]:
                t += s[0]
                gm.addIMUMeasurement(t, s[1:4], s[4:7])
            gm.reserveNode(t)
            for i in np.nonzero(seq.btw_b == k)[0]:
                gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), np.eye(6) * seq.btw_cov[i])
            gm.solve()
        outs.append(gm.trajectory(0, 60))
        gm.close()
    assert helpers.ate(outs[0], outs[1])[0] <= 1e-8
