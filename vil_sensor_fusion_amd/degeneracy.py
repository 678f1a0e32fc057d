"""Batched degeneracy metrics on the GPU (K6): the counterpart of the reference's
apply_degen_function (vil_fusion/python/make_prettier_graphs.py:547-576) and of the shipped
D-optimality gate (gtsam_fusion/src/degerate_odometry_filter.cpp:29-47)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

# order of degen_funcs (vil_fusion/python/degeneracy_detection_functions.py:283-303) + 2 extras
METRICS = ["d_opt", "d_opt_ratio", "a_opt", "a_opt_ratio", "e_opt", "e_opt_ratio", "max_eigen",
           "max_eigen_ratio", "jensen_bregman", "correlation_matrix_distance", "kullback_leibler",
           "norm_frobenius", "norm_frobenius_ratio", "norm_nuclear", "norm_nuclear_ratio", "norm_1",
           "norm_1_ratio", "norm_2", "norm_2_ratio", "condition_number", "differential_entropy"]
SUBSETS = {"all": 0, "trans": 1, "rot": 2}
ROT_DEGEN_THRESHOLD, TRANS_DEGEN_THRESHOLD = 11.5, 28.9     # config/carla/fusion_params.yaml:35-36


def apply_degen_function(matrix, pose, matrix_subset, func, dtype=np.float64, reps=0):
    """Same contract as the reference: matrix (6,6,T), pose (6,1,T) or None, subset in
    {"all","trans","rot"}, func = metric name (or a reference function object, matched by
    __name__).  Returns y (T,), y[0] = 0.  With reps > 0 also returns the kernel time in ms."""
    name = func if isinstance(func, str) else func.__name__
    if name not in METRICS:
        raise KeyError(f"metric {name!r} is not implemented on the GPU")
    if matrix_subset not in SUBSETS:
        raise RuntimeWarning("Invalid matrix subset {}".format(matrix_subset))   # as the reference (:560)
    m = np.ascontiguousarray(np.asarray(matrix).transpose(2, 0, 1), dtype=dtype)
    if m.shape[1:] != (6, 6):
        raise ValueError("matrix must be (6,6,T)")
    p = None
    if pose is not None:
        p = np.ascontiguousarray(np.asarray(pose)[:, 0, :].T, dtype=dtype)
    out = np.zeros(m.shape[0], dtype=dtype)
    ms = C.c_float(0)
    check(_lib.lib().vf_degeneracy_batch(
        m.ctypes.data_as(C.c_void_p), None if p is None else p.ctypes.data_as(C.c_void_p), m.shape[0],
        0 if dtype == np.float64 else 1, SUBSETS[matrix_subset], METRICS.index(name),
        out.ctypes.data_as(C.c_void_p), reps, C.byref(ms)))
    return (out, ms.value) if reps > 0 else out


def spectrum(matrix, matrix_subset="all", dtype=np.float64, reps=0):
    """e_opt, max_eigen and condition_number of every matrix of a (6,6,T) stack from ONE eigen-solve each (one launch, three
    outputs; bit for bit what three apply_degen_function calls return).  condition_number is NaN where a matrix is not
    symmetric to rounding.  Returns a dict (and the kernel time in ms with reps > 0)."""
    if matrix_subset not in SUBSETS:
        raise RuntimeWarning("Invalid matrix subset {}".format(matrix_subset))
    m = np.ascontiguousarray(np.asarray(matrix).transpose(2, 0, 1), dtype=dtype)
    if m.shape[1:] != (6, 6):
        raise ValueError("matrix must be (6,6,T)")
    outs = [np.zeros(m.shape[0], dtype=dtype) for _ in range(3)]
    ms = C.c_float(0)
    check(_lib.lib().vf_degeneracy_spectrum_batch(m.ctypes.data_as(C.c_void_p), m.shape[0], 0 if dtype == np.float64 else 1, SUBSETS[matrix_subset],
                                                 *[o.ctypes.data_as(C.c_void_p) for o in outs], reps, C.byref(ms)))
    res = dict(e_opt=outs[0], max_eigen=outs[1], condition_number=outs[2])
    return (res, ms.value) if reps > 0 else res


def dopt_filter(hessians, rot_thr=ROT_DEGEN_THRESHOLD, trans_thr=TRANS_DEGEN_THRESHOLD):
    """(T,36) or (T,6,6) float32 LOAM Hessians -> (rot_dopt, trans_dopt, keep) like the shipped node."""
    h = np.ascontiguousarray(hessians, dtype=np.float32).reshape(-1, 36)
    n = h.shape[0]
    rot, trans, keep = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8)
    check(_lib.lib().vf_dopt_filter_f32(h.ctypes.data_as(C.POINTER(C.c_float)), n, C.c_float(rot_thr),
                                        C.c_float(trans_thr), rot.ctypes.data_as(C.POINTER(C.c_float)),
                                        trans.ctypes.data_as(C.POINTER(C.c_float)),
                                        keep.ctypes.data_as(C.POINTER(C.c_ubyte))))
    return rot, trans, keep.astype(bool)
