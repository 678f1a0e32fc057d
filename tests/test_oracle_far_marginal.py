"""The linear algebra of k_marginalize<FAR> (far between factors marginalised with the keyframe that leaves a fixed-lag
window), restated in numpy (oracle/far_marginal.py) and checked against plain dense elimination: the marginal prior plus the
rows of the linear far factor ARE the Schur complement -- with several far factors at once, with a far end folded into the
prior's third keyframe, with every far end folded.  No GPU."""
import numpy as np
import pytest

from oracle import far_marginal as fm


def _problem(rng, T, R=None, scale=1.0):
    M = rng.normal(size=(60, 42))
    A = M.T @ M + np.eye(42) * 1e-3
    g = rng.normal(size=42)
    R = 6 * T if R is None else R
    W = rng.normal(size=(R, 42 + 6 * T)) * scale
    W[:, 21:30] = 0.0                       # the prior's second and third keyframes enter by their pose only (columns 30..35, 36..41)
    r = rng.normal(size=R)
    return A, g, W, r


@pytest.mark.parametrize("T,fold", [(1, ()), (3, ()), (3, (1,)), (2, (0, 1)), (8, (2, 5))])
def test_split_is_the_schur_complement(T, fold):
    rng = np.random.default_rng(100 + T)
    A, g, W, r = _problem(rng, T, scale=30.0)
    prior, eta, U, rn, live = fm.marginalize_with_far(A, g, W, r, fold)
    S, e = fm.dense_marginal(A, g, W, r, fold)
    n_live = len(live)
    assert U.shape == (6 * n_live, 27 + 6 * n_live) and live == [t for t in range(T) if t not in fold]
    J = np.zeros_like(S)
    J[:27, :27] = prior
    J += U.T @ U
    h = np.concatenate([eta, np.zeros(6 * n_live)]) + U.T @ rn
    assert np.abs(J - S).max() <= 1e-9 * np.abs(S).max()
    assert np.abs(h - e).max() <= 1e-9 * max(1.0, np.abs(e).max())
    # the prior on its own is a marginal (positive semi-definite), the far-end block of the rows is upper triangular
    assert np.linalg.eigvalsh(prior).min() >= -1e-9 * np.abs(prior).max()
    if n_live:
        assert np.abs(np.tril(U[:, 27:], -1)).max() == 0.0


def test_marginalising_a_shared_keyframe_couples_the_far_ends():
    """Why the window holds ONE linear far factor: two between factors m -> b1 and m -> b2 touch nothing but m and their own
    far end, yet once m is eliminated the marginal has a b1-b2 block -- which two separate six-row factors, each over the
    prior's keyframes and its own far end, cannot carry."""
    rng = np.random.default_rng(7)
    A, g, _, _ = _problem(rng, 2)
    W = np.zeros((12, 54))
    W[:6, :6], W[:6, 42:48] = rng.normal(size=(6, 6)) * 30, rng.normal(size=(6, 6)) * 30        # J_a on the pose of m, J_b on b1
    W[6:, :6], W[6:, 48:54] = rng.normal(size=(6, 6)) * 30, rng.normal(size=(6, 6)) * 30       # ... on b2
    r = rng.normal(size=12)
    S, _ = fm.dense_marginal(A, g, W, r)
    assert np.abs(S[27:33, 33:39]).max() > 1e-2 * np.abs(S[27:, 27:]).max()
    prior, eta, U, rn, live = fm.marginalize_with_far(A, g, W, r)
    assert np.abs(U[6:, 27:33]).max() == 0.0 and np.abs(U[:6, 33:39]).max() > 0.0                # rows of b1 reach b2 (L^T is upper triangular)
