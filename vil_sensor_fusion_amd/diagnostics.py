"""ROS-free core of the reference's diagnostics node (gtsam_fusion/python/diagnostics.py:33-141): every field of
gtsam_fusion/msg/DiagnosticMessage.msg:1-14 for one (ground truth, estimate) pair of frames.

The reference computes them from TF lookups; here the caller hands over the two poses in the stationary reference frame
(what TF holds), stamp by stamp.  Host-side evaluation of results (numpy); nothing here is on the hot path.
Quaternions are (w, x, y, z) like everywhere in this package (tf's are (x, y, z, w): rot_err[3] there is q[0] here)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .metrics import _qconj, _qmul, _qrot


@dataclass
class DiagnosticMessage:
    """gtsam_fusion/msg/DiagnosticMessage.msg, field for field (header.stamp -> stamp, err -> err_position / err_orientation)"""
    stamp: float = 0.0
    name: str = ""
    gt_distance: float = 0.0
    abs_dist_err: float = 0.0
    abs_rot_err: float = 0.0
    relative_dist_err: float = math.inf
    abs_linear_vel_err: float = 0.0
    abs_rot_vel_err: float = 0.0
    rel_linear_vel_err: float = math.inf
    rel_rot_vel_err: float = math.inf
    err_position: np.ndarray = field(default_factory=lambda: np.zeros(3))
    err_orientation: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.0, 0.0, 0.0]))   # w, x, y, z


def _angle(q):
    """2 acos |q_w| (diagnostics.py:114,127)"""
    return 2.0 * math.acos(min(1.0, abs(float(q[0]))))


def relative_transform(q_last, t_last, q_now, t_now):
    """lookupTransformFull(target_frame=f, target_time=last, source_frame=f, source_time=now, fixed_frame=ref)
    (diagnostics.py:84-101): the pose of frame f at `now` expressed in frame f at `last`, T_last^-1 T_now."""
    return _qmul(_qconj(q_last), q_now), _qrot(_qconj(q_last), np.asarray(t_now, float) - np.asarray(t_last, float))


class DiagnosticTrack:
    """One entry of the node's `diagnostics` parameter list (name, gt, est, ref, rate): feed update() the two poses at
    each common stamp; it returns the message the reference would publish (None for the first stamp, which only
    initialises last_time, diagnostics.py:67-71)."""

    def __init__(self, name: str):
        self.name = name
        self.total_distance = 0.0
        self._last = None

    def update(self, stamp, q_gt, t_gt, q_est, t_est):
        q_gt, q_est = np.asarray(q_gt, float), np.asarray(q_est, float)
        t_gt, t_est = np.asarray(t_gt, float), np.asarray(t_est, float)
        last, self._last = self._last, (q_gt, t_gt, q_est, t_est)
        if last is None:
            return None
        gt_d_rot, gt_d_trans = relative_transform(last[0], last[1], q_gt, t_gt)          # :84-91
        est_d_rot, est_d_trans = relative_transform(last[2], last[3], q_est, t_est)      # :94-101
        self.total_distance += float(np.linalg.norm(gt_d_trans))                         # :103
        lin_vel_diff = est_d_trans - gt_d_trans                                          # :105
        ang_vel_diff = _qmul(gt_d_rot, _qconj(est_d_rot))                                # :106
        # lookupTransform(target=gt, source=est, now): the estimate expressed in the ground-truth frame (:108-112)
        rot_err = _qmul(_qconj(q_gt), q_est)
        trans_err = _qrot(_qconj(q_gt), t_est - t_gt)
        m = DiagnosticMessage(stamp=float(stamp), name=self.name)
        m.gt_distance = self.total_distance
        m.abs_dist_err = float(np.linalg.norm(trans_err))
        m.abs_rot_err = _angle(rot_err)
        m.relative_dist_err = math.inf if m.gt_distance == 0 else m.abs_dist_err / m.gt_distance
        m.abs_linear_vel_err = float(np.linalg.norm(lin_vel_diff))
        m.abs_rot_vel_err = _angle(ang_vel_diff)
        gt_step = float(np.linalg.norm(gt_d_trans))
        m.rel_linear_vel_err = math.inf if gt_step == 0 else m.abs_linear_vel_err / gt_step
        gt_turn = _angle(gt_d_rot)
        m.rel_rot_vel_err = math.inf if gt_turn == 0 else m.abs_rot_vel_err / gt_turn
        m.err_position, m.err_orientation = trans_err, rot_err
        return m
