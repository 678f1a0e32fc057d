"""Python handle on the batch engine of libvilfusion.so (numpy in / numpy out).

Mirrors the C ABI one to one (include/vilfusion.h); all arithmetic happens on the GPU."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import check

IMU_RECORD, BTW_RECORD, PRIOR_RECORD = 190, 28, 31
MAX_FAR, MAX_FAR_LIMIT = 8, 32      # VF_MAX_EXTRA, VF_MAX_FAR_LIMIT
STAGES = {"linearize_imu": 1, "linearize_between": 2, "assemble": 3, "solve": 4, "retract": 5,
          "decide": 6, "assemble_idle": 7}
# GraphManager.cpp:27-31: pose (rad x3, m x3), velocity, bias prior sigmas
REFERENCE_PRIOR_SIGMAS = np.array([1e-6] * 3 + [5e-5] * 3 + [1e-5] * 3 + [1e-7] * 6)


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


@dataclass
class EngineOpts:
    windows: int = 1
    capacity: int = 1088
    bandwidth: int = 3
    device: int = 0
    gravity: tuple = (0.0, 0.0, -9.81)
    lambda0: float = 1e-5
    lambda_up: float = 10.0
    lambda_down: float = 10.0
    lambda_min: float = 1e-12
    lambda_max: float = 1e10
    chunks: int = 0          # K4 form: 0 = auto (<= 128 windows: partitioned solve), 1 = sweeps, P >= 2 = P chunks
    # solver-form switches (None = the library's default, vf_engine_default_tuning; include/vilfusion.h vf_engine_tuning)
    sweep_two_sided_max: int | None = None
    hybrid_threshold: int | None = None
    cold_start: bool = False
    use_hip_graph: bool = False
    accept_rel: float | None = None      # LM accept tolerance (None = the library's default 1e-9; 0 = strict decrease)
    solve_split_min: int | None = None   # one-wave sweeps: from this many windows on, forward sweep / back substitution as two kernels
    solve_assemble_min: int | None = None  # one-wave sweeps: from this many windows on, the forward sweep assembles H itself (no K3)
    solve_assemble_waves: int | None = None  # 1: one wave per window; 2: eliminator + assembler wave on one LDS image
    # refined solve (vf_engine_opts.refine_iterations): conjugate-gradient corrections through J after every solve.  None = the
    # library's default (-1: 12 corrections once a window is longer than refine_min_keyframes = 1536), 0 = never, N = always N
    refine_iterations: int | None = None
    refine_min_keyframes: int | None = None
    refine_rel_stop: float | None = None
    gauge_floor: float | None = None     # floor of the marginal prior's information about global translation / yaw (None = default 3e-4; 0 = off)
    hybrid_active_list: int | None = None  # hybrid solves: sweeps take their windows from the compacted list of active ones (None = default 1)
    far_batch_columns: int | None = None   # single-window engines: the Woodbury columns of far factors as one batched solve (None = default 1)
    far_big_forms: int | None = None       # engines made for > 8 far factors: their dense systems in device memory even with <= 8 alive (None = default 0)
    incremental: int | None = None         # isam_step re-eliminates only from the first keyframe that changed (None = default 0)
    wildfire: float | None = None          # ... and its back substitution stops once increments change by <= this (None = default 0: bitwise)
    min_model_fidelity: float | None = None  # > 0: GTSAM's LM accept rule (modelFidelity > this; 1e-3 there) instead of accept_rel (None = default 0)
    max_far_factors: int | None = None     # far between factors a window may hold (None = default VF_MAX_EXTRA = 8; at most 32)
    lm_excursion: int | None = None      # non-monotone LM: provisional cost-raising trials per excursion (None = default: 3 on refining engines)


class Engine:
    def __init__(self, opts: EngineOpts = EngineOpts()):
        self._l = _lib.lib()
        o = _lib.EngineOptsC()
        self._l.vf_engine_default_opts(C.byref(o))
        o.windows, o.capacity, o.bandwidth, o.device = opts.windows, opts.capacity, opts.bandwidth, opts.device
        o.gravity[:] = list(opts.gravity)
        o.lambda0, o.lambda_up, o.lambda_down = opts.lambda0, opts.lambda_up, opts.lambda_down
        o.lambda_min, o.lambda_max = opts.lambda_min, opts.lambda_max
        o.chunks = opts.chunks
        o.cold_start = int(opts.cold_start)
        for name in ("accept_rel", "refine_iterations", "refine_min_keyframes", "refine_rel_stop", "lm_excursion", "gauge_floor", "incremental", "wildfire", "min_model_fidelity", "max_far_factors"):
            if getattr(opts, name) is not None:
                setattr(o, name, getattr(opts, name))
        t = _lib.EngineTuningC()
        self._l.vf_engine_default_tuning(C.byref(t))
        t.use_hip_graph = int(opts.use_hip_graph)
        for name in ("sweep_two_sided_max", "hybrid_threshold", "solve_split_min", "solve_assemble_min", "solve_assemble_waves", "hybrid_active_list", "far_batch_columns", "far_big_forms"):
            if getattr(opts, name) is not None:
                setattr(t, name, getattr(opts, name))
        self._h = C.c_void_p()
        check(self._l.vf_engine_create_tuned(C.byref(o), C.byref(t), C.byref(self._h)))
        self.opts = opts
        self.capacity = (opts.capacity + 63) // 64 * 64
        self.windows = opts.windows

    def close(self):
        if getattr(self, "_h", None):
            self._l.vf_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- staging
    def set_range(self, window, lo, hi):
        check(self._l.vf_engine_set_range(self._h, window, lo, hi))

    def set_states(self, window, k0, states):
        s = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 16)
        check(self._l.vf_engine_set_states(self._h, window, k0, s.shape[0], _d(s)))

    def get_states(self, window, k0, n):
        s = np.zeros((n, 16))
        check(self._l.vf_engine_get_states(self._h, window, k0, n, _d(s)))
        return s

    def set_imu(self, window, k0, rec):
        r = np.ascontiguousarray(rec, dtype=np.float64).reshape(-1, IMU_RECORD)
        check(self._l.vf_engine_set_imu(self._h, window, k0, r.shape[0], _d(r)))

    def set_between(self, window, a, b, rec):
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        r = np.ascontiguousarray(rec, dtype=np.float64).reshape(-1, BTW_RECORD)
        assert a.size == b.size == r.shape[0]
        check(self._l.vf_engine_set_between(self._h, window, a.size, _i(a), _i(b), _d(r)))

    def set_extra_between(self, window, a, b, rec):
        """far between factors of a window (any pair of keyframes; vf_engine_set_extra_between): REPLACES the window's list"""
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        r = np.ascontiguousarray(rec, dtype=np.float64).reshape(-1, BTW_RECORD)
        assert a.size == b.size == r.shape[0]
        check(self._l.vf_engine_set_extra_between(self._h, window, a.size, _i(a), _i(b), _d(r)))

    def get_extra_between(self, window):
        """(a, b, records, made linear by a marginalisation or re-anchored by a slide without one so far, dropped without a
        marginalisation so far, absorbed into the marginal prior so far): the window's nonlinear far factors as they stand"""
        n, tr, en, ab = C.c_int(), C.c_long(), C.c_long(), C.c_long()
        a, b, r = np.zeros(MAX_FAR_LIMIT, dtype=np.int32), np.zeros(MAX_FAR_LIMIT, dtype=np.int32), np.zeros((MAX_FAR_LIMIT, BTW_RECORD))
        check(self._l.vf_engine_get_extra_between(self._h, window, C.byref(n), _i(a), _i(b), _d(r), C.byref(tr), C.byref(en), C.byref(ab)))
        return a[:n.value].copy(), b[:n.value].copy(), r[:n.value].copy(), tr.value, en.value, ab.value

    def get_linear_far(self, window):
        """window-local keyframes the window's LINEAR far factors end at (far factors whose older keyframe has been marginalised)"""
        n, b = C.c_int(), np.zeros(MAX_FAR_LIMIT, dtype=np.int32)
        check(self._l.vf_engine_get_linear_far(self._h, window, C.byref(n), _i(b)))
        return b[:n.value].copy()

    def clear_between(self, window, k0, n):
        check(self._l.vf_engine_clear_between(self._h, window, k0, n))

    def set_prior(self, window, k, rec):
        r = np.ascontiguousarray(rec, dtype=np.float64).reshape(PRIOR_RECORD)
        check(self._l.vf_engine_set_prior(self._h, window, k, _d(r)))

    def preintegrate(self, window, k0, step_off, steps, bias_hat, imu_cov):
        """K0 on the device: factors for keyframes k0..k0+n-1 from raw IMU steps (n = len(step_off)-1)."""
        off = np.ascontiguousarray(step_off, dtype=np.int32)
        st = np.ascontiguousarray(steps, dtype=np.float64).reshape(-1, 7)
        n = off.size - 1
        bh = np.ascontiguousarray(np.broadcast_to(np.asarray(bias_hat, dtype=np.float64), (n, 6)))
        p = _lib.ImuParamsC(imu_cov["acc"], imu_cov["gyro"], imu_cov["integration"], imu_cov["bias_acc"],
                            imu_cov["bias_omega"], imu_cov["bias_acc_omega_int"])
        check(self._l.vf_engine_preintegrate(self._h, window, k0, n, _i(off), _d(st), _d(bh), C.byref(p)))

    def ingest_tail(self, step_off, steps, imu_cov, btw_a, btw_rec):
        """the ingest half of a fixed-lag update for all windows (vf_engine_ingest_tail): window w's next IMU factor from its
        raw samples steps[step_off[w]:step_off[w+1]], preintegrated with the window's current bias estimate, + the between
        record ending at the new keyframe (btw_a[w] = source slot or -1).  Asynchronous."""
        off = np.ascontiguousarray(step_off, dtype=np.int32)
        st = np.ascontiguousarray(steps, dtype=np.float64).reshape(-1, 7)
        a = np.ascontiguousarray(btw_a, dtype=np.int32)
        r = np.ascontiguousarray(btw_rec, dtype=np.float64).reshape(-1, BTW_RECORD)
        assert off.size == self.windows + 1 and a.size == self.windows and r.shape[0] == self.windows
        p = _lib.ImuParamsC(imu_cov["acc"], imu_cov["gyro"], imu_cov["integration"], imu_cov["bias_acc"],
                            imu_cov["bias_omega"], imu_cov["bias_acc_omega_int"])
        check(self._l.vf_engine_ingest_tail(self._h, _i(off), _d(st), C.byref(p), _i(a), _d(r)))

    def ingest_status(self):
        """waits for the last ingest_tail, raises what its kernel reported, returns (h2d_ms, k0_ms) of that call"""
        h, k = C.c_float(), C.c_float()
        check(self._l.vf_engine_ingest_status(self._h, C.byref(h), C.byref(k)))
        return h.value, k.value

    def get_imu(self, window, k0, n):
        r = np.zeros((n, IMU_RECORD))
        check(self._l.vf_engine_get_imu(self._h, window, k0, n, _d(r)))
        return r

    # ---- stages
    def linearize(self, which=0):
        check(self._l.vf_engine_linearize(self._h, which))

    def assemble(self):
        check(self._l.vf_engine_assemble(self._h))

    def solve(self):
        check(self._l.vf_engine_solve(self._h))

    def retract(self):
        check(self._l.vf_engine_retract(self._h))

    def decide(self, init=False):
        check(self._l.vf_engine_decide(self._h, int(init)))

    def iterate(self, iterations):
        check(self._l.vf_engine_iterate(self._h, iterations))

    def slide(self, prior_sigma=REFERENCE_PRIOR_SIGMAS, marginalize=True):
        """Fixed-lag slide by one keyframe; marginalize=False re-anchors tight priors instead."""
        s = np.ascontiguousarray(prior_sigma, dtype=np.float64)
        check(self._l.vf_engine_slide(self._h, _d(s), int(marginalize)))

    def compact(self, shift):
        check(self._l.vf_engine_compact(self._h, shift))

    def grow(self, new_capacity):
        """more keyframe slots per window; states, factors and priors are carried over (vf_engine_grow)"""
        check(self._l.vf_engine_grow(self._h, new_capacity))
        self.capacity = (new_capacity + 63) // 64 * 64

    def marginalize(self):
        check(self._l.vf_engine_marginalize(self._h))

    def drop_oldest(self):
        check(self._l.vf_engine_drop_oldest(self._h))

    def read_marginal(self, window):
        on = C.c_int()
        x, L, eta = np.zeros((3, 16)), np.zeros((27, 27)), np.zeros(27)
        check(self._l.vf_engine_read_marginal(self._h, window, C.byref(on), _d(x), _d(L), _d(eta)))
        return dict(on=on.value, xbar=x, L=L, eta=eta)

    def predict(self, window, k0, n):
        """Initial values of keyframes [k0, k0+n) by IMU prediction from k0-1; window = -1: every window."""
        check(self._l.vf_engine_predict(self._h, window, k0, n))

    def isam_step(self, relin_threshold=1e-4):
        """One reference-compat update (vf_engine_isam_step): get_states = linearisation points, get_estimate = estimate."""
        check(self._l.vf_engine_isam_step(self._h, C.c_double(relin_threshold)))

    def incremental_info(self, window=0):
        """incremental engines: updates so far, how many of them eliminated the whole window, and for `window` the slot the
        last forward sweep started at / the last back substitution stopped at"""
        u, f, a, b = C.c_long(), C.c_long(), C.c_int(), C.c_int()
        check(self._l.vf_engine_incremental_info(self._h, window, C.byref(u), C.byref(f), C.byref(a), C.byref(b)))
        return dict(updates=u.value, whole_window_updates=f.value, first_eliminated=a.value, last_substituted=b.value)

    def gn_begin(self, relin_threshold=1e-4):
        """the opening of a reference-compat update on its own (time-sharded callers stage the solve themselves)"""
        check(self._l.vf_engine_gn_begin(self._h, C.c_double(relin_threshold)))

    # ---- refined solve (include/vilfusion.h "refine_iterations")
    def refine_count(self):
        n = C.c_int()
        check(self._l.vf_engine_refine_count(self._h, C.byref(n)))
        return n.value

    def refine_begin(self):
        check(self._l.vf_engine_refine_begin(self._h))

    def refine_step(self):
        check(self._l.vf_engine_refine_step(self._h))

    def refine_end(self):
        check(self._l.vf_engine_refine_end(self._h))

    def read_refine(self, window):
        """(corrections applied by the last refined solve, res . M^-1 res at its end / at its start)"""
        it, red = C.c_int(), C.c_double()
        check(self._l.vf_engine_read_refine(self._h, window, C.byref(it), C.byref(red)))
        return it.value, red.value

    def predict_from_estimate(self, window, k0, n):
        check(self._l.vf_engine_predict_from_estimate(self._h, window, k0, n))

    def get_estimate(self, window, k0, n):
        s = np.zeros((n, 16))
        check(self._l.vf_engine_get_estimate(self._h, window, k0, n, _d(s)))
        return s

    def sync(self):
        check(self._l.vf_engine_sync(self._h))

    def graph_info(self):
        """use_hip_graph: (replay active, captures of the launch sequence, iterate calls served by hipGraphLaunch)"""
        en, cap, rep = C.c_int(), C.c_int(), C.c_long()
        check(self._l.vf_engine_graph_info(self._h, C.byref(en), C.byref(cap), C.byref(rep)))
        return bool(en.value), cap.value, rep.value

    SOLVE_FORMS = ("one_wave", "one_wave_split", "assembling", "two_sided", "partitioned", "hybrid")

    def solve_form(self):
        """which form of K4 the next solve launches (vf_engine_solve_form)"""
        f = C.c_int()
        check(self._l.vf_engine_solve_form(self._h, C.byref(f)))
        return self.SOLVE_FORMS[f.value]

    # ---- time-sharded windows (see include/vilfusion.h; the collectives live in distributed.ShardedSolver)
    def set_stream(self, hip_stream):
        """Run every later stage on the caller's HIP stream (an integer hipStream_t, 0 = default stream)."""
        check(self._l.vf_engine_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_shard(self, rank, world):
        check(self._l.vf_engine_set_shard(self._h, rank, world))

    def shard_info(self):
        info = _lib.ShardInfoC()
        check(self._l.vf_engine_shard_info(self._h, C.byref(info)))
        return info

    def solve_local(self):
        check(self._l.vf_engine_solve_local(self._h))

    def solve_global(self):
        check(self._l.vf_engine_solve_global(self._h))

    def reset_lambda(self):
        check(self._l.vf_engine_reset_lambda(self._h))

    def set_async(self, on=True):
        """asynchronous staging (vf_engine_set_async; what the GraphManager's engine runs with): staging calls enqueue and return,
        device-detected failures travel in sticky words, vf_engine_read_result is the one synchronisation of a solve"""
        check(self._l.vf_engine_set_async(self._h, int(bool(on))))

    def read_result(self, window, slot, estimate=False):
        """vf_engine_read_result: state of keyframe `slot`, cost and LM counters of `window`, the sticky device flags -- one
        synchronisation (none when vf_engine_iterate has just left the block behind: vf_engine.hip "res_cached")"""
        st = np.zeros(16)
        cost, acc, rej, fl, fg = C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(self._l.vf_engine_read_result(self._h, window, slot, int(bool(estimate)), _d(st), C.byref(cost), C.byref(acc), C.byref(rej), C.byref(fl), C.byref(fg)))
        return dict(state=st, cost=cost.value, accepted=acc.value, rejected=rej.value, solve_failures=fl.value, device_flags=fg.value)

    def set_convergence(self, rel_tol=1e-5, abs_tol=1e-5):
        """Stop a window's LM trials once an accepted step lowers its cost by <= abs_tol or by <= rel_tol * cost
        (GTSAM's LM rule; off by default, (0, 0) switches it off)."""
        check(self._l.vf_engine_set_convergence(self._h, C.c_double(rel_tol), C.c_double(abs_tol)))

    # ---- read-back
    def read_imu_lin(self, window, k0, n, which=0):
        r, J = np.zeros((n, 15)), np.zeros((n, 15, 30))
        check(self._l.vf_engine_read_imu_lin(self._h, window, which, k0, n, _d(r), _d(J)))
        return r, J

    def read_between_lin(self, window, k0, n, which=0):
        r, Ja, Jb = np.zeros((n, 6)), np.zeros((n, 6, 6)), np.zeros((n, 6, 6))
        check(self._l.vf_engine_read_between_lin(self._h, window, which, k0, n, _d(r), _d(Ja), _d(Jb)))
        return r, Ja, Jb

    def read_normal(self, window, k0, n):
        H, g = np.zeros((n, 4, 15, 15)), np.zeros((n, 15))
        check(self._l.vf_engine_read_normal(self._h, window, k0, n, _d(H), _d(g)))
        return H, g

    def pose_information(self, window, k0, n):
        """(6, 6, n): the pose block of each keyframe's diagonal block of J^T J at the current
        linearisation (information of pose k given the rest of the window) -- the per-keyframe 6x6
        blocks the degeneracy metrics (K6) are run on in the batch pipeline (BASELINE configs[2])."""
        H, _ = self.read_normal(window, k0, n)
        return np.ascontiguousarray(H[:, 0, :6, :6].transpose(1, 2, 0))

    def read_delta(self, window, k0, n):
        d = np.zeros((n, 15))
        check(self._l.vf_engine_read_delta(self._h, window, k0, n, _d(d)))
        return d

    def read_panels(self, window, k0, n):
        p = np.zeros((n, 43, 16))
        check(self._l.vf_engine_read_panels(self._h, window, k0, n, _d(p)))
        return p

    def read_lm(self, window):
        cost, lam = C.c_double(), C.c_double()
        acc, rej, fails = C.c_int(), C.c_int(), C.c_int()
        check(self._l.vf_engine_read_lm(self._h, window, C.byref(cost), C.byref(lam), C.byref(acc),
                                        C.byref(rej), C.byref(fails)))
        return dict(cost=cost.value, lam=lam.value, accepted=acc.value, rejected=rej.value,
                    solve_failures=fails.value)

    def close_excursions(self):
        check(self._l.vf_engine_close_excursions(self._h))

    def read_excursions(self, window):
        """non-monotone LM: (trials kept provisionally so far, provisional trials of the excursion open now)"""
        a, b = C.c_int(), C.c_int()
        check(self._l.vf_engine_read_excursions(self._h, window, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- measurement
    def time_stage(self, stage, reps=10):
        ms = C.c_float()
        check(self._l.vf_engine_time_stage(self._h, STAGES[stage], reps, C.byref(ms)))
        return ms.value

    def time_iterate(self, iterations):
        ms = C.c_float()
        check(self._l.vf_engine_time_iterate(self._h, iterations, C.byref(ms)))
        return ms.value

    def counts(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(self._l.vf_engine_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(imu=a.value, between=b.value, keyframes=c.value)
