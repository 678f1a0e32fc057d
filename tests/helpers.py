"""Test-side construction of window problems (uses the CPU oracle; tests only)."""
import numpy as np

from vil_sensor_fusion_amd import synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS


def build_problem(oracle, seq, perturb=0.0, rng=None):
    """Factor records + initial values for one synthetic sequence.

    IMU factors are preintegrated with zero bias estimate (GraphManager's bias is zero until
    the first solve), initial values come from the IMU prediction chain
    (GraphManager.cpp:152-160), the prior sits on keyframe 0 with the reference's sigmas."""
    prm = oracle.carla_imu_params()
    n = seq.n
    recs = np.zeros((n, 190))
    for k in range(1, n):
        p = oracle.pim_new(np.zeros(6))
        for s in seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]:
            oracle.pim_integrate(p, prm, s[1:4], s[4:7], s[0])
        recs[k] = oracle.pim_to_record(p)
    g = np.array([0.0, 0.0, -9.81])
    states = np.zeros((n, 16))
    states[0] = seq.gt_states[0]
    for k in range(1, n):
        states[k] = oracle.predict(recs[k], g, states[k - 1])
    if perturb > 0:
        rng = rng or np.random.default_rng(seq.seed + 1000)
        for k in range(1, n):
            states[k] = oracle.retract(states[k], rng.normal(size=15) * perturb)
    return dict(n=n, states=states, imu=recs, btw_a=seq.btw_a, btw_b=seq.btw_b,
                btw=synth.between_records(seq),
                prior=synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS), gravity=g)


def oracle_window(oracle, prob, lo=0, hi=None):
    """oracle.Window over keyframes [lo, hi) of a problem (local indices shifted by lo)."""
    hi = prob["n"] if hi is None else hi
    m = (prob["btw_a"] >= lo) & (prob["btw_b"] < hi)
    ks = np.arange(lo + 1, hi)
    prior = prob["prior"]
    return oracle.Window(prob["states"][lo:hi], ks - 1 - lo, ks - lo, prob["imu"][lo + 1:hi],
                         prob["btw_a"][m] - lo, prob["btw_b"][m] - lo, prob["btw"][m],
                         np.array([0], dtype=np.int32), prior.reshape(1, -1), prob["gravity"])


def load_engine(eng, window, prob, lo=0, hi=None):
    hi = prob["n"] if hi is None else hi
    eng.set_states(window, 0, prob["states"])
    eng.set_imu(window, 1, prob["imu"][1:])
    eng.set_between(window, prob["btw_a"], prob["btw_b"], prob["btw"])
    eng.set_prior(window, lo, prob["prior"])
    eng.set_range(window, lo, hi)


def ate(states_a, states_b):
    """sqrt(mean |t_a - t_b|^2), no alignment (gauge fixed by the X0 prior); and the max
    rotation error 2 acos|q_w| of q_a^-1 q_b (gtsam_fusion/python/diagnostics.py:114)."""
    d = states_a[:, 4:7] - states_b[:, 4:7]
    qa, qb = states_a[:, :4], states_b[:, :4]
    w = np.abs(np.sum(qa * qb, axis=1)).clip(max=1.0)
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1)))), float(np.max(2 * np.arccos(w)))


SWEEP_TOLS = (1e-3, 1e-4, 1e-5, 1e-6, 1e-7)


def tolerance_sweep(y, ref, mask=None):
    """Fraction of entries of y within each relative tolerance of ref (entry 0 of a metric series is
    the reference's y[0] = 0 placeholder and is skipped); mask selects a stretch of the sequence."""
    y, ref = np.asarray(y, dtype=np.float64)[1:], np.asarray(ref, dtype=np.float64)[1:]
    if mask is not None:
        y, ref = y[mask[1:]], ref[mask[1:]]
    rel = np.abs(y - ref) / np.maximum(np.abs(ref), 1e-300)
    return [float(np.mean(rel <= t)) for t in SWEEP_TOLS]
