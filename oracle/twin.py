"""Independent numpy twin of the C oracle's conventions -- test infrastructure only.

Written from the mathematics, not from oracle/vf_oracle.c: rotations are 3x3 matrices, exponentials and logarithms are
scipy's generic matrix functions (expm / logm), the SO(3) right Jacobian is its power series, and every factor Jacobian
is a central difference of the twin's own residual.  What it shares with the C oracle (and the HIP kernels) is therefore
only the CONVENTIONS under test: tangent orders, retractions, the preintegration recursion, the residual definitions.
The 15x15 preintegrated covariance is propagated by `preintegrate_cov`: every Jacobian in it is a central difference of
the twin's own one-sample step (no closed-form A, B, C), composed in numpy; what it takes from GTSAM is the documented
SELECTION of blocks (which sensitivities enter F and which noises enter G, see the function).

State layout as everywhere: q(w,x,y,z) t(3) v(3) bias_acc(3) bias_gyro(3).
"""
import numpy as np
from scipy.linalg import expm, logm


def hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def vee(W):
    return np.array([W[2, 1], W[0, 2], W[1, 0]])


def so3_exp(w):
    return expm(hat(np.asarray(w, dtype=float)))


def so3_log(R):
    return vee(np.real(logm(R)))


def so3_jr(w, terms=30):
    """right Jacobian: sum_n (-1)^n / (n+1)! hat(w)^n"""
    W, J, P, f = hat(w), np.zeros((3, 3)), np.eye(3), 1.0
    for n in range(terms):
        f *= (n + 1)
        J += ((-1) ** n / f) * P
        P = P @ W
    return J


def quat_to_rot(q):
    w, x, y, z = np.asarray(q, dtype=float) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def se3_exp(xi):
    """tangent order [omega, v] -> (R, t), the full exponential map"""
    M = np.zeros((4, 4))
    M[:3, :3] = hat(xi[:3])
    M[:3, 3] = xi[3:]
    E = expm(M)
    return E[:3, :3], E[:3, 3]


def se3_log(R, t):
    M = np.eye(4)
    M[:3, :3], M[:3, 3] = R, t
    L = np.real(logm(M))
    return np.concatenate([vee(L[:3, :3]), L[:3, 3]])


def unpack(state):
    s = np.asarray(state, dtype=float)
    return quat_to_rot(s[:4]), s[4:7].copy(), s[7:10].copy(), s[10:16].copy()


# ---------------------------------------------------------------- preintegration (tangent form), mean only
def preintegrate(steps, bhat):
    """steps: rows (dt, acc xyz, gyro xyz).  Returns dt_ij and the 9-vector (theta, p, v):
    theta' = theta + Jr(theta)^-1 w dt ; p' = p + v dt + a_nav dt^2 / 2 ; v' = v + a_nav dt, a_nav = Exp(theta) a."""
    th, p, v, T = np.zeros(3), np.zeros(3), np.zeros(3), 0.0
    bhat = np.asarray(bhat, dtype=float)
    for s in np.atleast_2d(steps):
        dt, a, w = s[0], s[1:4] - bhat[:3], s[4:7] - bhat[3:]
        a_nav = so3_exp(th) @ a
        w_t = np.linalg.solve(so3_jr(th), w)
        th, p, v, T = th + w_t * dt, p + v * dt + 0.5 * a_nav * dt * dt, v + a_nav * dt, T + dt
    return T, np.concatenate([th, p, v])


def step(x, bias, meas, dt):
    """ONE sample of the recursion above on x = (theta, p, v) with the measurement corrected by `bias` (6)."""
    th, p, v = x[:3], x[3:6], x[6:9]
    a, w = meas[:3] - bias[:3], meas[3:] - bias[3:]
    a_nav = so3_exp(th) @ a
    w_t = np.linalg.solve(so3_jr(th), w)
    return np.concatenate([th + w_t * dt, p + v * dt + 0.5 * a_nav * dt * dt, v + a_nav * dt])


def _fd(f, x0, h):
    """central-difference Jacobian of f at x0"""
    y0 = f(x0)
    J = np.zeros((y0.size, x0.size))
    for c in range(x0.size):
        e = np.zeros(x0.size)
        e[c] = h
        J[:, c] = (f(x0 + e) - f(x0 - e)) / (2 * h)
    return J


def preintegrate_cov(steps, bhat, acc_cov, gyro_cov, int_cov, bias_acc_cov, bias_omega_cov, bias_acc_omega_int, h=1e-5):
    """15 x 15 covariance of (theta, p, v, bias_acc, bias_omega) after the samples in `steps`, by first-order propagation
    P <- F P F^T + Q per sample, every sensitivity a central difference of `step`:

        N_x = d step / d x      (9 x 9)        N_b = d step / d bias   (9 x 6; = - d step / d measurement)

    and the composition GTSAM 4.0.x documents for PreintegratedCombinedMeasurements::integrateMeasurement
    (called at IMUManager.cpp:50,64; "we consider the uncertainty of the bias selection and we keep correlation between
    biases and preintegrated measurements"):
      F = [[N_x, S], [0, I6]] with S holding ONLY theta/bias_omega (N_b[0:3, 3:6]) and v/bias_acc (N_b[6:9, 0:3]) -- the
          position/bias_acc sensitivity is left out there ("TODO: should we not also account for bias on position?");
      Q = blockdiag( Nt (gyro_cov + int_gg) Nt^T / dt,  dt int_cov,  Nv (acc_cov + int_aa) Nv^T / dt,
                     dt bias_acc_cov,  dt bias_omega_cov )   with Nt = N_b[0:3, 3:6], Nv = N_b[6:9, 0:3],
          int_aa / int_gg = the diagonal 3x3 blocks of biasAccOmegaInt (ImuManagerRos.cpp:28-33 sets it to c * I6, whose
          off-diagonal block -- the D_v_R term -- is zero).
    Discrete white noise of density c sampled at dt has covariance c / dt: hence N c N^T / dt with N proportional to dt."""
    x, bhat = np.zeros(9), np.asarray(bhat, dtype=float)
    P = np.zeros((15, 15))
    I3 = np.eye(3)
    for s in np.atleast_2d(steps):
        dt, meas = s[0], np.asarray(s[1:7], dtype=float)
        Nx = _fd(lambda y: step(y, bhat, meas, dt), x, h)
        Nb = _fd(lambda b: step(x, b, meas, dt), bhat, h)
        Nt, Nv = Nb[0:3, 3:6], Nb[6:9, 0:3]
        F = np.zeros((15, 15))
        F[:9, :9] = Nx
        F[0:3, 12:15] = Nt
        F[6:9, 9:12] = Nv
        F[9:, 9:] = np.eye(6)
        Q = np.zeros((15, 15))
        Q[0:3, 0:3] = Nt @ ((gyro_cov + bias_acc_omega_int) * I3) @ Nt.T / dt
        Q[3:6, 3:6] = dt * int_cov * I3
        Q[6:9, 6:9] = Nv @ ((acc_cov + bias_acc_omega_int) * I3) @ Nv.T / dt
        Q[9:12, 9:12] = dt * bias_acc_cov * I3
        Q[12:15, 12:15] = dt * bias_omega_cov * I3
        P = F @ P @ F.T + Q
        x = step(x, bhat, meas, dt)
    return P


def bias_jacobian_fd(steps, bhat, h=1e-6):
    """d preintegrated / d bias by central differences of the twin's own recursion (9 x 6)."""
    H = np.zeros((9, 6))
    for c in range(6):
        e = np.zeros(6)
        e[c] = h
        H[:, c] = (preintegrate(steps, np.asarray(bhat) + e)[1] - preintegrate(steps, np.asarray(bhat) - e)[1]) / (2 * h)
    return H


def corrected_delta(rec, bias_i):
    """biasCorrectedDelta: first-order correction of the preintegrated 9-vector for the current bias estimate"""
    d, bh, H = rec[1:10], rec[10:16], rec[16:70].reshape(9, 6)
    return d + H @ (np.asarray(bias_i) - bh)


def predict(rec, gravity, state_i):
    """NavState predicted at j: retract of state_i by (theta~, p~ + dt R_i^T v_i + dt^2/2 R_i^T g, v~ + dt R_i^T g)"""
    Ri, ti, vi, bi = unpack(state_i)
    dt = rec[0]
    x = corrected_delta(rec, bi)
    g = np.asarray(gravity, dtype=float)
    xp = x[3:6] + dt * (Ri.T @ vi) + 0.5 * dt * dt * (Ri.T @ g)
    xv = x[6:9] + dt * (Ri.T @ g)
    return Ri @ so3_exp(x[:3]), ti + Ri @ xp, vi + Ri @ xv


def imu_residual(rec, gravity, state_i, state_j):
    """unwhitened CombinedImuFactor residual (15): localCoordinates(state_j -> predicted) and bias_i - bias_j"""
    Rp, tp, vp = predict(rec, gravity, state_i)
    Rj, tj, vj, bj = unpack(state_j)
    bi = np.asarray(state_i, dtype=float)[10:16]
    return np.concatenate([so3_log(Rj.T @ Rp), Rj.T @ (tp - tj), Rj.T @ (vp - vj), bi - bj])


def _perturb(state, kind, d):
    """kind 'pose': (R Exp(d[:3]), t + R d[3:]) ; 'vel': v + d (navigation frame) ; 'bias': bias + d"""
    s = np.array(state, dtype=float)
    if kind == "pose":
        R = quat_to_rot(s[:4]) @ so3_exp(d[:3])
        s[4:7] = s[4:7] + quat_to_rot(s[:4]) @ d[3:]
        s[:4] = rot_to_quat(R)
    elif kind == "vel":
        s[7:10] += d
    else:
        s[10:16] += d
    return s


def rot_to_quat(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    q = np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
    return q / np.linalg.norm(q)


def imu_jacobian_fd(rec, gravity, state_i, state_j, h=1e-6):
    """15 x 30 in GTSAM key order X_i(6) V_i(3) X_j(6) V_j(3) B_i(6) B_j(6), central differences"""
    cols = [("i", "pose", 6), ("i", "vel", 3), ("j", "pose", 6), ("j", "vel", 3), ("i", "bias", 6), ("j", "bias", 6)]
    J, c0 = np.zeros((15, 30)), 0
    for who, kind, n in cols:
        for c in range(n):
            d = np.zeros(n)
            d[c] = h
            if who == "i":
                rp = imu_residual(rec, gravity, _perturb(state_i, kind, d), state_j)
                rm = imu_residual(rec, gravity, _perturb(state_i, kind, -d), state_j)
            else:
                rp = imu_residual(rec, gravity, state_i, _perturb(state_j, kind, d))
                rm = imu_residual(rec, gravity, state_i, _perturb(state_j, kind, -d))
            J[:, c0 + c] = (rp - rm) / (2 * h)
        c0 += n
    return J


def between_residual(rec, state_a, state_b):
    """unwhitened BetweenFactor<Pose3> residual (6, [rot, trans]): Logmap(measured^-1 (T_a^-1 T_b))"""
    Ra, ta, _, _ = unpack(state_a)
    Rb, tb, _, _ = unpack(state_b)
    Rm, tm = quat_to_rot(rec[:4]), np.asarray(rec[4:7], dtype=float)
    Rh, th = Ra.T @ Rb, Ra.T @ (tb - ta)
    return se3_log(Rm.T @ Rh, Rm.T @ (th - tm))


def between_jacobian_fd(rec, state_a, state_b, h=1e-6):
    Ja, Jb = np.zeros((6, 6)), np.zeros((6, 6))
    for c in range(6):
        d = np.zeros(6)
        d[c] = h
        Ja[:, c] = (between_residual(rec, _perturb(state_a, "pose", d), state_b) -
                    between_residual(rec, _perturb(state_a, "pose", -d), state_b)) / (2 * h)
        Jb[:, c] = (between_residual(rec, state_a, _perturb(state_b, "pose", d)) -
                    between_residual(rec, state_a, _perturb(state_b, "pose", -d))) / (2 * h)
    return Ja, Jb
