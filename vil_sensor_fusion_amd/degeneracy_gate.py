"""The shipped LiDAR degeneracy gate (gtsam_fusion/src/degerate_odometry_filter.cpp:29-47) over the C ABI: float32 log det
of the rotation (3,3) and translation (0,0) 3x3 blocks of the 36-float scan-matching Hessian; an odometry message is
republished only when neither is below its threshold (config/carla/fusion_params.yaml:35-36).  The arithmetic runs on
the GPU (vf_dopt_filter_f32); single messages and whole bags go through the same call."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class DegeneracyGate:
    def __init__(self, rot_degen_threshold=11.5, trans_degen_threshold=28.9):
        self.rot_thr, self.trans_thr = float(rot_degen_threshold), float(trans_degen_threshold)
        self.dropped = 0

    def evaluate(self, hessians36):
        """(n, 36) float32 -> keep (n,) bool, rot_dopt (n,), trans_dopt (n,)"""
        h = np.ascontiguousarray(hessians36, dtype=np.float32).reshape(-1, 36)
        n = h.shape[0]
        rot, trans, keep = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        check(_lib.lib().vf_dopt_filter_f32(fp(h), n, C.c_float(self.rot_thr), C.c_float(self.trans_thr), fp(rot), fp(trans),
                                            keep.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return keep.astype(bool), rot, trans

    def __call__(self, hessian36) -> bool:
        """one synchronised (odometry, OptStatus) pair: True = republish the odometry message"""
        keep, _, _ = self.evaluate(np.asarray(hessian36, dtype=np.float32).reshape(1, 36))
        if not keep[0]:
            self.dropped += 1
        return bool(keep[0])
