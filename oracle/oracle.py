"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It builds oracle/build/libvf_oracle.so on demand (gcc) and exposes the C
restatement through numpy arrays.  Nothing under vil_sensor_fusion_amd/ imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libvf_oracle.so")

IMU_DATA = 190
BTW_DATA = 28
PRIOR_DATA = 31


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, "vf_oracle.c"), os.path.join(_HERE, "vf_oracle.h")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class ImuParams(C.Structure):
    _fields_ = [("acc_cov", C.c_double), ("gyro_cov", C.c_double), ("int_cov", C.c_double),
                ("bias_acc_cov", C.c_double), ("bias_omega_cov", C.c_double),
                ("bias_acc_omega_int", C.c_double), ("gravity", C.c_double * 3)]


class Pim(C.Structure):
    _fields_ = [("dt", C.c_double), ("d", C.c_double * 9), ("bhat", C.c_double * 6),
                ("H", C.c_double * 54), ("cov", C.c_double * 225)]


class Marg(C.Structure):
    _fields_ = [("on", C.c_int), ("k0", C.c_int), ("xbar", C.c_double * 48), ("L", C.c_double * 729),
                ("eta", C.c_double * 27)]

    def arrays(self):
        return dict(on=self.on, k0=self.k0, xbar=np.array(self.xbar[:]).reshape(3, 16),
                    L=np.array(self.L[:]).reshape(27, 27), eta=np.array(self.eta[:]))


class Problem(C.Structure):
    _fields_ = [("n_kf", C.c_int), ("states", C.POINTER(C.c_double)),
                ("n_imu", C.c_int), ("imu_i", C.POINTER(C.c_int32)), ("imu_j", C.POINTER(C.c_int32)),
                ("imu_data", C.POINTER(C.c_double)),
                ("n_btw", C.c_int), ("btw_a", C.POINTER(C.c_int32)), ("btw_b", C.POINTER(C.c_int32)),
                ("btw_data", C.POINTER(C.c_double)),
                ("n_prior", C.c_int), ("prior_k", C.POINTER(C.c_int32)),
                ("prior_data", C.POINTER(C.c_double)),
                ("gravity", C.c_double * 3), ("marg", C.POINTER(Marg))]


GAUGE_FLOOR = 3e-4   # default of vf_engine_opts.gauge_floor: the floor of the WINDOW's softest eigenvalues


def prior_gauge_floor(n_kf, floor=GAUGE_FLOOR):
    """what k_marginalize lifts the marginal prior's gauge information to, for a window of n_kf keyframes (floor * n / 3)"""
    return floor * n_kf / 3.0

ACCEPT_REL = 1e-9    # default accept tolerance of vfo_lm = vf_engine_opts.accept_rel's default (include/vilfusion.h).  A constant:
#                      callers that want another rule pass accept_rel to Window.lm / FixedLagOracle, nobody assigns to this name


class LmOpts(C.Structure):
    _fields_ = [("lambda0", C.c_double), ("lambda_up", C.c_double), ("lambda_down", C.c_double),
                ("lambda_min", C.c_double), ("lambda_max", C.c_double),
                ("iterations", C.c_int), ("n_threads", C.c_int),
                ("rel_tol", C.c_double), ("abs_tol", C.c_double), ("accept_rel", C.c_double),
                ("refine", C.c_int), ("refine_rel_stop", C.c_double), ("excursion", C.c_int), ("min_model_fidelity", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.vfo_cost.restype = C.c_double
        _lib.vfo_assemble.restype = C.c_double
        _lib.vfo_lm.restype = C.c_double
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _arr(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def carla_imu_params() -> ImuParams:
    """gtsam_fusion/config/carla/fusion_params.yaml:22-27, MakeSharedU gravity."""
    p = ImuParams()
    p.bias_acc_cov = 1e-4
    p.bias_omega_cov = 1e-6
    p.acc_cov = 1e-6
    p.gyro_cov = 1e-6
    p.int_cov = 1e-8
    p.bias_acc_omega_int = 1e-4
    p.gravity[:] = [0.0, 0.0, -9.81]
    return p


def make_imu_params(acc, gyro, integ, bias_acc, bias_omega, bias_int, gravity=(0, 0, -9.81)):
    p = ImuParams()
    p.acc_cov, p.gyro_cov, p.int_cov = acc, gyro, integ
    p.bias_acc_cov, p.bias_omega_cov, p.bias_acc_omega_int = bias_acc, bias_omega, bias_int
    p.gravity[:] = list(gravity)
    return p


# ---------------------------------------------------------------- Lie group primitives
def so3_exp(w):
    R = np.zeros(9)
    lib().vfo_so3_exp(_d(_arr(w, 3)), _d(R))
    return R.reshape(3, 3)


def so3_log(R):
    w = np.zeros(3)
    lib().vfo_so3_log(_d(_arr(R, 9)), _d(w))
    return w


def so3_jr(w):
    J = np.zeros(9)
    lib().vfo_so3_jr(_d(_arr(w, 3)), _d(J))
    return J.reshape(3, 3)


def so3_jr_inv(w):
    J = np.zeros(9)
    lib().vfo_so3_jr_inv(_d(_arr(w, 3)), _d(J))
    return J.reshape(3, 3)


def quat_to_rot(q):
    R = np.zeros(9)
    lib().vfo_quat_to_rot(_d(_arr(q, 4)), _d(R))
    return R.reshape(3, 3)


def rot_to_quat(R):
    q = np.zeros(4)
    lib().vfo_rot_to_quat(_d(_arr(R, 9)), _d(q))
    return q


def se3_exp(xi):
    R, t = np.zeros(9), np.zeros(3)
    lib().vfo_se3_exp(_d(_arr(xi, 6)), _d(R), _d(t))
    return R.reshape(3, 3), t


def se3_log(R, t):
    xi = np.zeros(6)
    lib().vfo_se3_log(_d(_arr(R, 9)), _d(_arr(t, 3)), _d(xi))
    return xi


def se3_jr_inv(xi):
    J = np.zeros(36)
    lib().vfo_se3_jr_inv(_d(_arr(xi, 6)), _d(J))
    return J.reshape(6, 6)


# ---------------------------------------------------------------- preintegration
def pim_new(bhat=np.zeros(6)) -> Pim:
    p = Pim()
    lib().vfo_pim_reset(C.byref(p), _d(_arr(bhat, 6)))
    return p


def pim_integrate(p: Pim, prm: ImuParams, acc, gyro, dt):
    lib().vfo_pim_integrate(C.byref(p), C.byref(prm), _d(_arr(acc, 3)), _d(_arr(gyro, 3)),
                            C.c_double(dt))


def imu_get_factor(t, acc, gyro, head, start, end, bias, prm):
    """IMUManager::getFactor on a sorted buffer; returns (pim, new_head, n_integrations)."""
    t = _arr(t)
    acc = _arr(acc, 3 * t.size)
    gyro = _arr(gyro, 3 * t.size)
    h = C.c_int(head)
    out = Pim()
    n = lib().vfo_imu_get_factor(_d(t), _d(acc), _d(gyro), C.c_int(t.size), C.byref(h),
                                 C.c_double(start), C.c_double(end), _d(_arr(bias, 6)),
                                 C.byref(prm), C.byref(out))
    return out, h.value, n


def sqrt_info_upper(cov):
    cov = _arr(cov)
    n = int(round(np.sqrt(cov.size)))
    Rp = np.zeros(n * (n + 1) // 2)
    rc = lib().vfo_sqrt_info_upper(_d(cov), C.c_int(n), _d(Rp))
    if rc != 0:
        raise np.linalg.LinAlgError("covariance not SPD")
    return Rp


def unpack_upper(Rp, n):
    R = np.zeros((n, n))
    R[np.triu_indices(n)] = Rp
    return R


def pim_to_record(p: Pim):
    rec = np.zeros(IMU_DATA)
    rc = lib().vfo_pim_to_record(C.byref(p), _d(rec))
    if rc != 0:
        raise np.linalg.LinAlgError("preintMeasCov not SPD")
    return rec


def pim_fields(p: Pim):
    return dict(dt=p.dt, d=np.array(p.d[:]), bhat=np.array(p.bhat[:]),
                H=np.array(p.H[:]).reshape(9, 6), cov=np.array(p.cov[:]).reshape(15, 15))


def predict(rec, gravity, state_i):
    out = np.zeros(16)
    lib().vfo_predict(_d(_arr(rec, IMU_DATA)), _d(_arr(gravity, 3)), _d(_arr(state_i, 16)), _d(out))
    return out


# ---------------------------------------------------------------- factors
def imu_factor(rec, gravity, xi, xj, whiten=True):
    r, J = np.zeros(15), np.zeros(450)
    lib().vfo_imu_factor(_d(_arr(rec, IMU_DATA)), _d(_arr(gravity, 3)), _d(_arr(xi, 16)),
                         _d(_arr(xj, 16)), C.c_int(int(whiten)), _d(r), _d(J))
    return r, J.reshape(15, 30)


def between_factor(rec, xa, xb, whiten=True):
    r, Ja, Jb = np.zeros(6), np.zeros(36), np.zeros(36)
    lib().vfo_between_factor(_d(_arr(rec, BTW_DATA)), _d(_arr(xa, 16)), _d(_arr(xb, 16)),
                             C.c_int(int(whiten)), _d(r), _d(Ja), _d(Jb))
    return r, Ja.reshape(6, 6), Jb.reshape(6, 6)


def prior_factor(rec, x):
    r, J = np.zeros(15), np.zeros(225)
    lib().vfo_prior_factor(_d(_arr(rec, PRIOR_DATA)), _d(_arr(x, 16)), _d(r), _d(J))
    return r, J.reshape(15, 15)


def retract(x, delta):
    out = np.zeros(16)
    lib().vfo_retract(_d(_arr(x, 16)), _d(_arr(delta, 15)), _d(out))
    return out


# ---------------------------------------------------------------- window problem
class Window:
    """Owns numpy buffers for one window problem and the matching C struct."""

    def __init__(self, states, imu_i, imu_j, imu_data, btw_a, btw_b, btw_data, prior_k,
                 prior_data, gravity=(0.0, 0.0, -9.81)):
        self.states = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 16).copy()
        self.imu_i = np.ascontiguousarray(imu_i, dtype=np.int32)
        self.imu_j = np.ascontiguousarray(imu_j, dtype=np.int32)
        self.imu_data = np.ascontiguousarray(imu_data, dtype=np.float64).reshape(-1, IMU_DATA)
        self.btw_a = np.ascontiguousarray(btw_a, dtype=np.int32)
        self.btw_b = np.ascontiguousarray(btw_b, dtype=np.int32)
        self.btw_data = np.ascontiguousarray(btw_data, dtype=np.float64).reshape(-1, BTW_DATA)
        self.prior_k = np.ascontiguousarray(prior_k, dtype=np.int32)
        self.prior_data = np.ascontiguousarray(prior_data, dtype=np.float64).reshape(-1, PRIOR_DATA)
        p = Problem()
        p.n_kf = self.states.shape[0]
        p.states = _d(self.states)
        p.n_imu = self.imu_i.size
        p.imu_i, p.imu_j, p.imu_data = _i(self.imu_i), _i(self.imu_j), _d(self.imu_data)
        p.n_btw = self.btw_a.size
        p.btw_a, p.btw_b, p.btw_data = _i(self.btw_a), _i(self.btw_b), _d(self.btw_data)
        p.n_prior = self.prior_k.size
        p.prior_k, p.prior_data = _i(self.prior_k), _d(self.prior_data)
        p.gravity[:] = list(gravity)
        p.marg = None
        self.marg = None
        self.c = p

    @property
    def n_kf(self):
        return self.states.shape[0]

    def set_marg(self, marg):
        """attach a marginal prior (oracle.Marg, k0 window-local) or None"""
        self.marg = marg
        self.c.marg = C.pointer(marg) if marg is not None else None

    def marginalize(self, m=0, gauge_floor=0.0):
        """Schur complement of every factor touching keyframe m onto [m+1:15][m+2:pose][m+3:pose]; gauge_floor:
        the floor of the PRIOR's gauge information, prior_gauge_floor(n) for the engine's behaviour (the oracle does what it is told)."""
        out = Marg()
        rc = lib().vfo_marginalize_floor(C.byref(self.c), C.c_int(m), C.c_double(gauge_floor), C.byref(out))
        if rc != 0:
            raise np.linalg.LinAlgError("marginalisation failed")
        return out

    def cost(self):
        return lib().vfo_cost(C.byref(self.c))

    def bandwidth(self):
        return lib().vfo_bandwidth(C.byref(self.c))

    def assemble(self, w=None, n_threads=1):
        w = self.bandwidth() if w is None else w
        H = np.zeros((self.n_kf, w + 1, 15, 15))
        g = np.zeros((self.n_kf, 15))
        cost = lib().vfo_assemble(C.byref(self.c), C.c_int(w), _d(H), _d(g), C.c_int(n_threads))
        return cost, H, g

    def lm(self, iterations=5, lambda0=1e-5, up=10.0, down=10.0, lmin=1e-12, lmax=1e10,
           n_threads=1, rel_tol=0.0, abs_tol=0.0, accept_rel=None, refine=0, refine_rel_stop=1e-8, excursion=0, min_model_fidelity=0.0):
        """refine / excursion: the engine's refined solve and non-monotone accept rule (vf_engine_opts.refine_iterations,
        lm_excursion); 0 / 0 = the classical normal-equation LM.  The engine switches both on by itself for windows longer
        than 1536 keyframes; the oracle does what it is told."""
        o = LmOpts(lambda0, up, down, lmin, lmax, iterations, n_threads, rel_tol, abs_tol, ACCEPT_REL if accept_rel is None else accept_rel,
                   refine, refine_rel_stop, excursion, min_model_fidelity)
        costs = np.zeros(iterations + 1)
        acc = np.zeros(iterations, dtype=np.int32)
        lam = lib().vfo_lm(C.byref(self.c), C.byref(o), _d(costs),
                           acc.ctypes.data_as(C.POINTER(C.c_int)))
        return costs, acc, lam


def gn_step(win: "Window", refine=0, refine_rel_stop=1e-8):
    """one undamped Gauss-Newton update of win.states (vfo_gn_step); returns (cost before, corrections applied)"""
    c, it = C.c_double(), C.c_int()
    rc = lib().vfo_gn_step(C.byref(win.c), C.c_int(refine), C.c_double(refine_rel_stop), C.byref(c), C.byref(it))
    if rc != 0:
        raise np.linalg.LinAlgError("normal equations not positive definite")
    return c.value, it.value


def band_solve(H, g, lam):
    n_kf, wp1 = H.shape[0], H.shape[1]
    d = np.zeros((n_kf, 15))
    rc = lib().vfo_band_solve(C.c_int(n_kf), C.c_int(wp1 - 1), _d(np.ascontiguousarray(H)),
                              _d(np.ascontiguousarray(g)), C.c_double(lam), _d(d))
    return rc, d
