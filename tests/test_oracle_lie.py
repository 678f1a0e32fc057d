"""Pins the oracle's SO(3)/SE(3) primitives against independent implementations
(scipy expm/logm, scipy Rotation) and its Jacobians against central differences."""
import numpy as np
import pytest
from scipy.linalg import expm, logm
from scipy.spatial.transform import Rotation


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def hat6(xi):
    T = np.zeros((4, 4))
    T[:3, :3] = hat(xi[:3])
    T[:3, 3] = xi[3:]
    return T


ANGLES = [0.0, 1e-9, 1e-5, 1e-3, 0.05, 0.4, 0.51, 1.3, 2.9]


@pytest.mark.parametrize("ang", ANGLES)
def test_so3_exp_log_vs_scipy(oracle, ang):
    rng = np.random.default_rng(1)
    ax = rng.normal(size=3)
    w = ax / np.linalg.norm(ax) * ang
    R = oracle.so3_exp(w)
    np.testing.assert_allclose(R, expm(hat(w)), atol=2e-15)
    np.testing.assert_allclose(R, Rotation.from_rotvec(w).as_matrix(), atol=2e-15)
    np.testing.assert_allclose(oracle.so3_log(R), w, atol=1e-15 + 2e-15 * ang)
    q = oracle.rot_to_quat(R)
    np.testing.assert_allclose(oracle.quat_to_rot(q), R, atol=2e-15)


@pytest.mark.parametrize("ang", ANGLES)
def test_se3_exp_log_vs_scipy(oracle, ang):
    rng = np.random.default_rng(2)
    ax = rng.normal(size=3)
    xi = np.concatenate([ax / np.linalg.norm(ax) * ang, rng.normal(size=3) * 2.0])
    R, t = oracle.se3_exp(xi)
    T = expm(hat6(xi))
    np.testing.assert_allclose(R, T[:3, :3], atol=3e-15)
    np.testing.assert_allclose(t, T[:3, 3], atol=1e-14)
    np.testing.assert_allclose(oracle.se3_log(R, t), xi, atol=2e-14)
    if ang > 1e-3:
        L = np.real(logm(T))
        np.testing.assert_allclose(oracle.se3_log(R, t)[:3], [L[2, 1], L[0, 2], L[1, 0]], atol=1e-12)
        np.testing.assert_allclose(oracle.se3_log(R, t)[3:], L[:3, 3], atol=1e-12)


@pytest.mark.parametrize("ang", [1e-6, 1e-3, 0.1, 0.49, 0.52, 1.7])
def test_so3_jacobians_fd(oracle, ang):
    rng = np.random.default_rng(3)
    ax = rng.normal(size=3)
    w = ax / np.linalg.norm(ax) * ang
    h = 1e-6
    # Exp(w + d) = Exp(w) Exp(Jr d)
    Jn = np.zeros((3, 3))
    for i in range(3):
        e = np.zeros(3); e[i] = h
        Jn[:, i] = (oracle.so3_log(oracle.so3_exp(w).T @ oracle.so3_exp(w + e))
                    - oracle.so3_log(oracle.so3_exp(w).T @ oracle.so3_exp(w - e))) / (2 * h)
    np.testing.assert_allclose(oracle.so3_jr(w), Jn, atol=1e-9)
    np.testing.assert_allclose(oracle.so3_jr_inv(w) @ oracle.so3_jr(w), np.eye(3), atol=1e-13)


@pytest.mark.parametrize("ang", [1e-7, 1e-3, 0.1, 0.49, 0.52, 1.7])
def test_se3_logmap_derivative_fd(oracle, ang):
    """Log(T Exp(d)) = Log(T) + Jr^{-1} d"""
    rng = np.random.default_rng(4)
    ax = rng.normal(size=3)
    xi = np.concatenate([ax / np.linalg.norm(ax) * ang, rng.normal(size=3)])
    R, t = oracle.se3_exp(xi)
    h = 1e-6
    Jn = np.zeros((6, 6))
    for i in range(6):
        e = np.zeros(6); e[i] = h
        dR, dt = oracle.se3_exp(e)
        p = oracle.se3_log(R @ dR, t + R @ dt)
        dR, dt = oracle.se3_exp(-e)
        m = oracle.se3_log(R @ dR, t + R @ dt)
        Jn[:, i] = (p - m) / (2 * h)
    np.testing.assert_allclose(oracle.se3_jr_inv(xi), Jn, atol=2e-9)


def test_se3_jr_inv_series(oracle):
    """Independent check: J_r^{-1}(xi) = sum_n B_n (-1)^n/n! ad^n for small xi."""
    rng = np.random.default_rng(5)
    xi = rng.normal(size=6) * 0.2
    W, V = hat(xi[:3]), hat(xi[3:])
    ad = np.zeros((6, 6))
    ad[:3, :3] = W; ad[3:, 3:] = W; ad[3:, :3] = V
    bern = [1, 0.5, 1 / 6, 0, -1 / 30, 0, 1 / 42, 0, -1 / 30, 0, 5 / 66, 0, -691 / 2730]  # B1=+1/2 (right)
    J = np.zeros((6, 6)); P = np.eye(6); f = 1.0
    for n, b in enumerate(bern):
        if n > 0:
            P = P @ ad; f *= n
        J += b / f * P
    np.testing.assert_allclose(oracle.se3_jr_inv(xi), J, atol=1e-13)
