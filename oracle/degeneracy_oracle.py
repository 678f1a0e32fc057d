"""CPU ORACLE for the degeneracy metrics (test infrastructure, NOT product code).

Batched-numpy restatement of the reference's scalar metrics
(vil_fusion/python/degeneracy_detection_functions.py:38-251) as called per message by
apply_degen_function (vil_fusion/python/make_prettier_graphs.py:547-576), plus the shipped
float32 D-optimality filter (gtsam_fusion/src/degerate_odometry_filter.cpp:29-47).

PINNED: tests/test_oracle_degeneracy.py checks every function here against
tests/golden/degeneracy_golden.npz, which was produced by importing the reference's own Python
in the build container (tests/golden/make_degeneracy_golden.py).
"""
from __future__ import annotations

import math

import numpy as np

METRICS = ["d_opt", "d_opt_ratio", "a_opt", "a_opt_ratio", "e_opt", "e_opt_ratio", "max_eigen",
           "max_eigen_ratio", "jensen_bregman", "correlation_matrix_distance", "kullback_leibler",
           "norm_frobenius", "norm_frobenius_ratio", "norm_nuclear", "norm_nuclear_ratio", "norm_1",
           "norm_1_ratio", "norm_2", "norm_2_ratio", "condition_number", "differential_entropy"]
SUBSETS = {"all": slice(0, 6), "trans": slice(0, 3), "rot": slice(3, 6)}


def subset(mats, pose, name):
    """make_prettier_graphs.py:548-560. mats (T,6,6), pose (T,6)."""
    s = SUBSETS[name]
    return mats[:, s, s], (None if pose is None else pose[:, s])


def _ratio(now, prev):
    return now @ np.linalg.inv(prev)


def _corr(m):
    # degeneracy_detection_functions.py:28-35: `*` is ELEMENTWISE on ndarrays, so only the
    # diagonal survives: diag_i = (1/sqrt(m_ii)) * m_ii * (1/sqrt(m_ii))
    d = np.sqrt(np.diagonal(m, axis1=-2, axis2=-1))
    dinv = 1.0 / d
    out = np.zeros_like(m)
    idx = np.arange(m.shape[-1])
    out[..., idx, idx] = dinv * np.diagonal(m, axis1=-2, axis2=-1) * dinv
    return out


def evaluate(metric: str, mats: np.ndarray, pose: np.ndarray | None = None) -> np.ndarray:
    """y[0] = 0, y[i] = metric(mat_now = mats[i], mat_prev = mats[i-1], pose_now, pose_prev)
    (make_prettier_graphs.py:562-576). mats (T,n,n), pose (T,n)."""
    T, n = mats.shape[0], mats.shape[1]
    y = np.zeros(T)
    if T < 2:
        return y
    now, prev = mats[1:], mats[:-1]
    with np.errstate(all="ignore"):
        if metric == "d_opt":                                   # :38-44
            y[1:] = np.exp(np.linalg.slogdet(now)[1] / n)
        elif metric == "d_opt_ratio":                           # :47-53
            y[1:] = np.exp(np.linalg.slogdet(_ratio(now, prev))[1] / n)
        elif metric == "a_opt":                                 # :56-60
            y[1:] = np.trace(now, axis1=1, axis2=2)
        elif metric == "a_opt_ratio":                           # :63-71
            y[1:] = np.trace(_ratio(now, prev), axis1=1, axis2=2)
        elif metric == "e_opt":                                 # :74-83
            y[1:] = np.real(np.linalg.eigvals(now)).min(axis=1)
        elif metric == "e_opt_ratio":                           # :86-96
            y[1:] = np.real(np.linalg.eigvals(_ratio(now, prev))).min(axis=1)
        elif metric == "max_eigen":                             # :99-108
            y[1:] = np.real(np.linalg.eigvals(now)).max(axis=1)
        elif metric == "max_eigen_ratio":                       # :111-120
            y[1:] = np.real(np.linalg.eigvals(_ratio(now, prev))).max(axis=1)
        elif metric == "jensen_bregman":                        # :123-129
            y[1:] = np.linalg.slogdet((now + prev) / 2)[1] - 0.5 * np.linalg.det(now @ prev)
        elif metric == "correlation_matrix_distance":           # :132-145
            a, b = _corr(now), _corr(prev)
            tr = np.trace(a @ b, axis1=1, axis2=2)
            y[1:] = 1 - tr / (np.linalg.norm(a, axis=(1, 2)) * np.linalg.norm(b, axis=(1, 2)))
        elif metric == "kullback_leibler":                      # :148-181  (E1 = prev, E2 = now)
            e2i = np.linalg.inv(now)
            a = np.trace(e2i @ prev - np.eye(n), axis1=1, axis2=2)
            du = (pose[:-1] - pose[1:])[..., None]
            b = (np.swapaxes(du, 1, 2) @ e2i @ du)[:, 0, 0]
            c = np.log(np.abs(np.linalg.det(now)) / np.abs(np.linalg.det(prev)))
            y[1:] = 0.5 * (a + b + c)
        elif metric == "norm_frobenius":                        # :203-204
            y[1:] = np.linalg.norm(now, axis=(1, 2))
        elif metric == "norm_frobenius_ratio":                  # :219-221
            y[1:] = np.linalg.norm(_ratio(now, prev), axis=(1, 2))
        elif metric == "norm_nuclear":                          # :207-208
            y[1:] = np.linalg.svd(now, compute_uv=False).sum(axis=1)
        elif metric == "norm_nuclear_ratio":                    # :224-226
            y[1:] = np.linalg.svd(_ratio(now, prev), compute_uv=False).sum(axis=1)
        elif metric == "norm_1":                                # :211-212
            y[1:] = np.abs(now).sum(axis=1).max(axis=1)
        elif metric == "norm_1_ratio":                          # :229-231
            y[1:] = np.abs(_ratio(now, prev)).sum(axis=1).max(axis=1)
        elif metric == "norm_2":                                # :215-216
            y[1:] = np.linalg.svd(now, compute_uv=False).max(axis=1)
        elif metric == "norm_2_ratio":                          # :234-236
            y[1:] = np.linalg.svd(_ratio(now, prev), compute_uv=False).max(axis=1)
        elif metric == "condition_number":                      # :239-243
            s = np.linalg.svd(now, compute_uv=False)
            y[1:] = -(s.max(axis=1) / s.min(axis=1))
        elif metric == "differential_entropy":                  # :195-200
            x = (2 * math.pi * math.e) ** n
            d = x * np.linalg.det(now)
            y[1:] = np.where(d > 0, 0.5 * np.log(np.where(d > 0, d, 1.0)), np.nan)
        else:
            raise KeyError(metric)
    return y


def dopt_filter_f32(hessians: np.ndarray, rot_thr: float, trans_thr: float):
    """degerate_odometry_filter.cpp:29-47 in float32: the 36 floats are copied into a
    column-major Eigen matrix (a transpose of the row-major message, :30-31), rotation block
    (3,3), translation block (0,0), log(det); keep iff neither is below its threshold."""
    h = np.asarray(hessians, dtype=np.float32).reshape(-1, 6, 6).transpose(0, 2, 1)

    def det3(a):  # Eigen's 3x3 determinant, evaluated in float32
        f = np.float32
        return (a[:, 0, 0] * (a[:, 1, 1] * a[:, 2, 2] - a[:, 1, 2] * a[:, 2, 1])
                - a[:, 0, 1] * (a[:, 1, 0] * a[:, 2, 2] - a[:, 1, 2] * a[:, 2, 0])
                + a[:, 0, 2] * (a[:, 1, 0] * a[:, 2, 1] - a[:, 1, 1] * a[:, 2, 0])).astype(f)
    with np.errstate(all="ignore"):
        rot = np.log(det3(h[:, 3:6, 3:6])).astype(np.float32)
        trans = np.log(det3(h[:, 0:3, 0:3])).astype(np.float32)
    keep = ~((rot < np.float32(rot_thr)) | (trans < np.float32(trans_thr)))
    return rot, trans, keep
