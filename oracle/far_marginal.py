"""CPU ORACLE for the marginalisation of far between factors (test infrastructure, NOT product code).

numpy restatement of what `k_marginalize<FAR>` (vil_sensor_fusion_amd/csrc/vf_kernels.hip) does when the keyframe m that
leaves a fixed-lag window is touched by far factors -- the device's own algorithm, not the reference's: the reference keeps
every BetweenFactor for good in an unbounded iSAM2 graph (gtsam_fusion/src/gtsam_fusion/GraphManager.cpp:83-88) and has no
fixed lag, so there is nothing of it to restate here; what pins the result is the whole-history optimum (the reference's
answer), which tests/test_gpu_far_factors.py compares a fixed-lag run with.  This file states the linear algebra on its own,
so that it can be checked without a GPU (tests/test_oracle_far_marginal.py): the split it produces IS the Schur complement.

Variables: m (the leaving keyframe, 15 dof), n (the three keyframes of the new marginal prior, 27 dof: [15][pose 6][pose 6]),
far ends b_1 .. b_T (pose, 6 dof each).  Inputs: A (42 x 42) and g (42), the information and gradient of everything else that
touches m (band factors, the old prior), over (m, n); W (R x (42 + 6 T)) and r (R), the whitened rows of the far factors over
(m, n, b) and their residuals at the current linearisation; fold: the far ends that have come within the prior's reach (their
keyframe is the third of n: columns 36..41 of the 42).
"""
from __future__ import annotations

import numpy as np


def marginalize_with_far(A, g, W, r, fold=()):
    """-> (prior_L (27 x 27), prior_eta (27), U (6 T' x (27 + 6 T')), r_new (6 T'), live): the new marginal prior on n and the
    rows of the linear far factor over (n, the live far ends), T' = far ends not folded, `live` their indices.

    1. fold: the columns of a far end at m + 3 join the pose columns of that keyframe (36..41);
    2. the joint information over (m, n, b): [[A + W_mn^T W_mn, W_mn^T W_b], [W_b^T W_mn, W_b^T W_b]], gradient [g + W_mn^T r; W_b^T r];
    3. eliminate m (15 pivots): the exact marginal S, eta over (n, b);
    4. split: S_bb = L L^T, X = L^-1 [S_bn | eta_b]; rows U = [X_n | L^T], residual x_eta; prior S_nn - X_n^T X_n, eta_n - X_n^T x_eta."""
    A, g, W, r = (np.asarray(x, dtype=np.float64) for x in (A, g, W, r))
    T = (W.shape[1] - 42) // 6
    W = W.copy()
    for t in fold:
        W[:, 36:42] += W[:, 42 + 6 * t:48 + 6 * t]
        W[:, 42 + 6 * t:48 + 6 * t] = 0.0
    live = [t for t in range(T) if t not in set(fold)]
    cols = list(range(42)) + [42 + 6 * t + c for t in live for c in range(6)]
    Wl = W[:, cols]
    N = len(cols)
    J = np.zeros((N, N))
    J[:42, :42] = A
    J += Wl.T @ Wl
    h = np.concatenate([g, np.zeros(N - 42)]) + Wl.T @ r
    # Schur complement of the leading 15 x 15 block
    Jmm, Jmr = J[:15, :15], J[:15, 15:]
    S = J[15:, 15:] - Jmr.T @ np.linalg.solve(Jmm, Jmr)
    eta = h[15:] - Jmr.T @ np.linalg.solve(Jmm, h[:15])
    if not live:
        return 0.5 * (S + S.T), eta, np.zeros((0, 27)), np.zeros(0), live
    Snn, Sbn, Sbb = S[:27, :27], S[27:, :27], S[27:, 27:]
    L = np.linalg.cholesky(0.5 * (Sbb + Sbb.T))
    X = np.linalg.solve(L, np.hstack([Sbn, eta[27:, None]]))
    Xn, xe = X[:, :27], X[:, 27]
    U = np.hstack([Xn, L.T])
    prior = Snn - Xn.T @ Xn
    return 0.5 * (prior + prior.T), eta[:27] - Xn.T @ xe, U, xe, live


def dense_marginal(A, g, W, r, fold=()):
    """The same marginal over (n, live far ends) by plain dense algebra on the stacked problem (for the test): information and
    gradient of 0.5 x^T A x + g^T x + 0.5 |W x + r|^2 with m eliminated."""
    A, g, W, r = (np.asarray(x, dtype=np.float64) for x in (A, g, W, r))
    T = (W.shape[1] - 42) // 6
    N = 42 + 6 * T
    J = np.zeros((N, N))
    J[:42, :42] = A
    J += W.T @ W
    h = np.concatenate([g, np.zeros(N - 42)]) + W.T @ r
    # a folded far end IS the keyframe at columns 36..41: identify the two variables (x_b = x_36..41)
    P = np.eye(N)
    keep = list(range(N))
    for t in fold:
        for c in range(6):
            P[42 + 6 * t + c, 36 + c] = 1.0
            keep.remove(42 + 6 * t + c)
    P = P[:, keep]
    J, h = P.T @ J @ P, P.T @ h
    S = J[15:, 15:] - J[:15, 15:].T @ np.linalg.solve(J[:15, :15], J[:15, 15:])
    eta = h[15:] - J[:15, 15:].T @ np.linalg.solve(J[:15, :15], h[:15])
    return S, eta
