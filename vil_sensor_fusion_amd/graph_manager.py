"""Python mirror of VILFusion::GraphManager / IMUManager over the C ABI.

Method names, argument meaning and error behaviour follow the reference
(gtsam_fusion/include/gtsam_fusion/GraphManager.h:40-96, IMUManager.h:22-27) so that the
reference's own test timelines (gtsam_fusion/test/UnitTests.cpp) read the same here.  Poses are
(q_wxyz, t) tuples instead of gtsam::Pose3; noise is a 6x6 covariance in Pose3 tangent order
[rot, trans] (what SensorManagerRos.cpp:99 wraps into noiseModel::Gaussian::Covariance).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

# gtsam_fusion/config/carla/fusion_params.yaml:22-27
CARLA_IMU = dict(acc=1e-6, gyro=1e-6, integration=1e-8, bias_acc=1e-4, bias_omega=1e-6,
                 bias_acc_omega_int=1e-4)


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class GraphManager:
    """GraphManager(imuManager): the IMU manager is folded in (addIMUMeasurement)."""

    def __init__(self, imu_params=CARLA_IMU, capacity=4096, lag=0, iterations=5, device=0,
                 prior_sigma=None, rel_tol=None, abs_tol=None, reference_compat=False, relin_threshold=None,
                 cold_start=False, fixed_capacity=False, incremental=False, wildfire=None, min_model_fidelity=None,
                 synchronous_staging=False, max_far_factors=None):
        """iterations: LM trials per solve at most; a solve stops earlier once a trial changes the cost by <= abs_tol or
        <= rel_tol * cost (defaults 1e-5 / 1e-5, gtsam::LevenbergMarquardtParams; 0 / 0: always `iterations` trials)."""
        self._l = _lib.lib()
        p = _lib.ImuParamsC(imu_params["acc"], imu_params["gyro"], imu_params["integration"],
                            imu_params["bias_acc"], imu_params["bias_omega"],
                            imu_params["bias_acc_omega_int"])
        o = _lib.GraphOptsC()
        self._l.vf_graph_default_opts(C.byref(o))
        o.capacity, o.lag, o.iterations, o.device = capacity, lag, iterations, device
        if prior_sigma is not None:
            o.prior_sigma[:] = list(prior_sigma)
        if rel_tol is not None:
            o.rel_tol = rel_tol
        if abs_tol is not None:
            o.abs_tol = abs_tol
        # reference_compat: solve() = ONE iSAM2-like update (relinearizeThreshold 1e-4, GraphManager.cpp:38-43), lag must be 0
        o.reference_compat = int(bool(reference_compat))
        o.cold_start = int(bool(cold_start))      # every solve linearises all factors (tests compare it with the warm start)
        o.fixed_capacity = int(bool(fixed_capacity))   # lag = 0: fail with VF_ERR_CAPACITY instead of growing the engine
        if relin_threshold is not None:
            o.relin_threshold = relin_threshold
        # incremental (with reference_compat): a solve re-eliminates only from the first keyframe that changed (vilfusion.h)
        o.incremental = int(incremental)          # (2: the incremental kernels over the whole history every time -- what tests compare with)
        if wildfire is not None:
            o.wildfire = wildfire
        o.synchronous_staging = int(bool(synchronous_staging))     # the pre-round-6 staging: same bits, slower (tests compare the two)
        if min_model_fidelity is not None:        # GTSAM's LM accept rule (LevenbergMarquardtParams::minModelFidelity = 1e-3) instead of the library's own
            o.min_model_fidelity = min_model_fidelity
        if max_far_factors is not None:           # loop closures alive at once (default = the limit, VF_MAX_FAR_LIMIT = 32)
            o.max_far_factors = max_far_factors
        self._h = C.c_void_p()
        check(self._l.vf_create(C.byref(p), C.byref(o), C.byref(self._h)))
        self._cbs = []

    def close(self):
        if getattr(self, "_h", None):
            self._l.vf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setInitialState(self, state16):
        """extra: anchor X(0), V(0), B(0) (and the means of their priors) at state16 = q t v bias instead of
        identity / zero; only before the first reserveNode (for logs that do not start level and at rest)."""
        s = np.ascontiguousarray(state16, dtype=np.float64).reshape(16)
        check(self._l.vf_set_initial_state(self._h, _d(s)))

    # IMUManager::addIMUMeasurement
    def addIMUMeasurement(self, time, accel, gyro):
        a = np.ascontiguousarray(accel, dtype=np.float64)
        w = np.ascontiguousarray(gyro, dtype=np.float64)
        check(self._l.vf_add_imu(self._h, C.c_double(time), _d(a), _d(w)))

    def reserveNode(self, time) -> int:
        key = C.c_uint64()
        check(self._l.vf_reserve_node(self._h, C.c_double(time), C.byref(key)))
        return key.value

    def getMostRecentPoseTime(self):
        t, k = C.c_double(), C.c_uint64()
        check(self._l.vf_most_recent_pose_time(self._h, C.byref(t), C.byref(k)))
        return t.value, k.value

    def addBetweenFactor(self, previousKey, currentKey, betweenPose, noiseCovariance):
        q = np.ascontiguousarray(betweenPose[0], dtype=np.float64)
        t = np.ascontiguousarray(betweenPose[1], dtype=np.float64)
        cov = np.ascontiguousarray(noiseCovariance, dtype=np.float64).reshape(6, 6)
        check(self._l.vf_add_between(self._h, C.c_uint64(previousKey), C.c_uint64(currentKey), _d(q),
                                     _d(t), _d(cov)))

    def addFactor(self, key, record190):
        """GraphManager::addFactor(const CombinedImuFactor&): queue a ready-made preintegrated factor ending at `key`
        (its 190-double record) instead of cutting one from the IMU buffer."""
        r = np.ascontiguousarray(record190, dtype=np.float64).reshape(190)
        check(self._l.vf_add_imu_factor(self._h, C.c_uint64(key), _d(r)))

    def getMostRecentEstimate(self):
        """GraphManager::getMostRecentEstimate: (q_wxyz, t, v) of a member the reference never assigns (identity, zero)."""
        q, t, v = np.zeros(4), np.zeros(3), np.zeros(3)
        check(self._l.vf_get_most_recent_estimate(self._h, _d(q), _d(t), _d(v)))
        return (q, t), v

    def solve(self):
        check(self._l.vf_solve(self._h))

    def addOptimizationCallback(self, callback):
        """callback(time, q_wxyz, t, v, bias) -- GraphManager::OptimizationCallback"""
        def tramp(_user, time, q, t, v, b):
            callback(time, np.array(q[:4]), np.array(t[:3]), np.array(v[:3]), np.array(b[:6]))
        cb = _lib.CALLBACK(tramp)
        self._cbs.append(cb)
        check(self._l.vf_set_callback(self._h, cb, None))

    def getState(self):
        q, t, v, b = np.zeros(4), np.zeros(3), np.zeros(3), np.zeros(6)
        check(self._l.vf_get_state(self._h, _d(q), _d(t), _d(v), _d(b)))
        return (q, t), v, b

    def getBias(self):
        b = np.zeros(6)
        check(self._l.vf_get_bias(self._h, _d(b)))
        return b

    def graphSize(self):
        """graph()->size(): factors staged since the last solve (3 priors initially)."""
        s, q = C.c_int(), C.c_int()
        check(self._l.vf_graph_staged(self._h, C.byref(s), C.byref(q)))
        return s.value

    def imuQueueSize(self):
        s, q = C.c_int(), C.c_int()
        check(self._l.vf_graph_staged(self._h, C.byref(s), C.byref(q)))
        return q.value

    def lmStats(self):
        """cost after the last solve; LM trials accepted / rejected / failed since creation (diagnostics)."""
        c = C.c_double()
        a, r, f = C.c_int(), C.c_int(), C.c_int()
        check(self._l.vf_graph_lm_stats(self._h, C.byref(c), C.byref(a), C.byref(r), C.byref(f)))
        return dict(cost=c.value, accepted=a.value, rejected=r.value, solve_failures=f.value)

    def solverInfo(self):
        """(keyframes in the window of the last solve, refinement corrections per solve now, provisional LM trials so far)"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self._l.vf_graph_solver_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def graph(self):
        """GraphManager::graph() (GraphManager.cpp:46-49): the staged factors as a list of dicts -- kind ("prior_pose" |
        "prior_velocity" | "prior_bias" | "between"), keys, measured (q_wxyz, t), covariance (6 x 6)."""
        n = self.graphSize()
        names = ("prior_pose", "prior_velocity", "prior_bias", "between")
        out = []
        for i in range(n):
            kind, k1, k2 = C.c_int(), C.c_uint64(), C.c_uint64()
            q, t, cov = np.zeros(4), np.zeros(3), np.zeros((6, 6))
            check(self._l.vf_graph_get_staged(self._h, i, C.byref(kind), C.byref(k1), C.byref(k2), _d(q), _d(t), _d(cov)))
            out.append(dict(kind=names[kind.value], keys=(k1.value, k2.value), measured=(q, t), covariance=cov))
        return out

    def incrementalInfo(self):
        """incremental handles: updates so far, how many re-eliminated the whole history, the keys the last update's forward
        sweep started at / its back substitution stopped at"""
        u, f, a, b = C.c_long(), C.c_long(), C.c_uint64(), C.c_uint64()
        check(self._l.vf_graph_incremental_info(self._h, C.byref(u), C.byref(f), C.byref(a), C.byref(b)))
        return dict(updates=u.value, whole_window_updates=f.value, first_eliminated_key=a.value, last_substituted_key=b.value)

    def trajectory(self, key0, n):
        s = np.zeros((n, 16))
        check(self._l.vf_get_trajectory(self._h, C.c_uint64(key0), n, _d(s)))
        return s

    def imuFactor(self, key):
        r = np.zeros(190)
        check(self._l.vf_get_imu_factor(self._h, C.c_uint64(key), _d(r)))
        return r
