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


def ate_gauge_aligned(states_a, states_b):
    """ATE after removing what no factor of a fixed-lag window can see: a global translation and a rotation about gravity (z)
    of trajectory a, fitted in closed form (planar Procrustes on x, y; mean offset in z).  For comparing two runs of a
    smoother whose global pose is carried by a decaying prior only."""
    pa, pb = states_a[:, 4:7], states_b[:, 4:7]
    ca, cb = pa.mean(axis=0), pb.mean(axis=0)
    xa, xb = pa - ca, pb - cb
    s = np.sum(xa[:, 0] * xb[:, 1] - xa[:, 1] * xb[:, 0])
    c = np.sum(xa[:, 0] * xb[:, 0] + xa[:, 1] * xb[:, 1])
    psi = np.arctan2(s, c)
    R = np.array([[np.cos(psi), -np.sin(psi), 0.0], [np.sin(psi), np.cos(psi), 0.0], [0.0, 0.0, 1.0]])
    d = xa @ R.T - xb
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1)))), float(psi), float(np.linalg.norm(cb - ca))


SWEEP_TOLS = (1e-3, 1e-4, 1e-5, 1e-6, 1e-7)


def tolerance_sweep(y, ref, mask=None):
    """Fraction of entries of y within each relative tolerance of ref (entry 0 of a metric series is
    the reference's y[0] = 0 placeholder and is skipped); mask selects a stretch of the sequence."""
    y, ref = np.asarray(y, dtype=np.float64)[1:], np.asarray(ref, dtype=np.float64)[1:]
    if mask is not None:
        y, ref = y[mask[1:]], ref[mask[1:]]
    rel = np.abs(y - ref) / np.maximum(np.abs(ref), 1e-300)
    return [float(np.mean(rel <= t)) for t in SWEEP_TOLS]


class FixedLagOracle:
    """The CPU oracle doing bench.py's fixed-lag update (Engine.slide(marginalize=True) + iterate(K)) on one window:

        init:    window [0, n) with the anchor prior of keyframe 0, K LM trials
        update:  marginalise the oldest keyframe at the current linearisation (vfo_marginalize: Schur complement of its
                 prior / previous marginal prior, its IMU factor and the between factors that start at it, onto
                 [next: 15][next+1: pose][next+2: pose]); append one keyframe, initial value by IMU prediction
                 (GraphManager.cpp:152-160); K LM trials on [s, s+n) with the marginal prior

    prob: helpers.build_problem(...) of a sequence with at least n + updates keyframes (its `states` beyond the first
    window are ignored: appended keyframes are predicted from the running estimate, as the engine does)."""

    def __init__(self, oracle, prob, n, iterations, threads=1, init_iterations=None, accept_rel=None, ingest=None, refine=0, gauge_floor=None, excursion=0):
        """init_iterations: LM trials of the initial solve (default: `iterations`).  A 1000-pose window started from
        IMU dead reckoning needs 50-150 trials to converge; slid while still far from its optimum it stays in a regime
        where two float64 implementations drift apart by 1e-5 m (DESIGN.md "Converged start")."""
        self.o, self.prob, self.n, self.K, self.threads = oracle, dict(prob, imu=prob["imu"].copy()), n, iterations, threads
        # ingest = (synth.Sequence, oracle.ImuParams): every update preintegrates the appended keyframe's IMU factor afresh
        # from its raw samples WITH THE CURRENT BIAS ESTIMATE (the bias of the window's last keyframe), as
        # GraphManager::reserveNode does (GraphManager.cpp:59) and as vf_engine_ingest_tail does on the device
        self.ingest = ingest
        self.accept_rel = accept_rel                 # None: the oracle's default (= the engine's); 0: strict decrease
        self.gauge_floor = oracle.GAUGE_FLOOR if gauge_floor is None else gauge_floor     # vf_engine_opts.gauge_floor (None: its default)
        self.excursion = excursion                   # vf_engine_opts.lm_excursion (with it lambda persists from solve to solve, as on the device)
        self.lam = None
        self.refine = refine                         # > 0: every solve refined through J (vf_engine_opts.refine_iterations)
        self.rel_tol = self.abs_tol = 0.0            # > 0: GTSAM's LM termination rule in the updates that follow
        self.states = prob["states"].copy()
        self.s, self.marg = 0, None
        self.win = self._window(0, None, True)
        self.costs, self.acc, lam = self.win.lm(iterations=iterations if init_iterations is None else init_iterations, n_threads=threads,
                                                accept_rel=accept_rel, refine=refine, excursion=excursion)
        self.lam = lam if excursion else None
        self.states[:n] = self.win.states

    def _window(self, lo, marg, with_prior):
        p, hi = self.prob, lo + self.n
        m = (p["btw_a"] >= lo) & (p["btw_b"] < hi)
        ks = np.arange(lo + 1, hi)
        pk = np.array([0], dtype=np.int32) if with_prior else np.zeros(0, dtype=np.int32)
        pd = p["prior"].reshape(1, -1) if with_prior else np.zeros((0, 31))
        w = self.o.Window(self.states[lo:hi], ks - 1 - lo, ks - lo, p["imu"][lo + 1:hi], p["btw_a"][m] - lo,
                          p["btw_b"][m] - lo, p["btw"][m], pk, pd, p["gravity"])
        if marg is not None:
            w.set_marg(marg)
        return w

    def update(self):
        """one fixed-lag update; returns the window's states afterwards (keyframes [s, s + n))"""
        p, n = self.prob, self.n
        self.marg = self.win.marginalize(0, self.o.prior_gauge_floor(self.n, self.gauge_floor))   # (self.win still holds the previous window, its prior / marginal prior attached)
        self.marg.k0 = 0
        self.s += 1
        s = self.s
        if self.ingest is not None:
            seq, prm = self.ingest
            new = s + n - 1
            pim = self.o.pim_new(self.states[new - 1, 10:16])
            for st in seq.imu_steps[seq.imu_off[new]:seq.imu_off[new + 1]]:
                self.o.pim_integrate(pim, prm, st[1:4], st[4:7], st[0])
            p["imu"][new] = self.o.pim_to_record(pim)
        self.states[s + n - 1] = self.o.predict(p["imu"][s + n - 1], p["gravity"], self.states[s + n - 2])
        self.win = self._window(s, self.marg, False)
        kw = {} if self.lam is None else {"lambda0": self.lam}
        self.costs, self.acc, lam = self.win.lm(iterations=self.K, n_threads=self.threads, rel_tol=self.rel_tol, abs_tol=self.abs_tol,
                                                accept_rel=self.accept_rel, refine=self.refine, excursion=self.excursion, **kw)
        self.lam = lam if self.excursion else None
        self.states[s:s + n] = self.win.states
        return self.win.states

    @property
    def trials(self):
        """LM trials the last update took (acc = -1 marks the ones a termination rule left out)"""
        return int(np.sum(np.asarray(self.acc) >= 0))

    @property
    def window_states(self):
        return self.states[self.s:self.s + self.n]
