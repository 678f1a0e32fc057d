"""Trajectory error definitions of the reference's diagnostics node
(gtsam_fusion/python/diagnostics.py:103-129), ROS/TF-free, plus the ATE used by BASELINE.json.

Host-side evaluation of results (numpy); nothing here is on the hot path."""
from __future__ import annotations

import numpy as np


def _qmul(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.stack([a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1] - a[..., 2] * b[..., 2] - a[..., 3] * b[..., 3],
                     a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0] + a[..., 2] * b[..., 3] - a[..., 3] * b[..., 2],
                     a[..., 0] * b[..., 2] - a[..., 1] * b[..., 3] + a[..., 2] * b[..., 0] + a[..., 3] * b[..., 1],
                     a[..., 0] * b[..., 3] + a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1] + a[..., 3] * b[..., 0]], axis=-1)


def _qconj(q):
    q = np.asarray(q, float)
    return q * np.array([1.0, -1.0, -1.0, -1.0])


def _qrot(q, v):
    qv = np.concatenate([np.zeros(np.shape(v)[:-1] + (1,)), np.asarray(v, float)], axis=-1)
    return _qmul(_qmul(q, qv), _qconj(q))[..., 1:]


def pose_error(q_gt, t_gt, q_est, t_est):
    """lookupTransform(target=gt, source=est) (diagnostics.py:107-111): est expressed in gt."""
    q_err = _qmul(_qconj(q_gt), q_est)
    t_err = _qrot(_qconj(q_gt), np.asarray(t_est, float) - np.asarray(t_gt, float))
    return q_err, t_err


def abs_dist_err(q_gt, t_gt, q_est, t_est):
    """msg.abs_dist_err = ||trans_err|| (diagnostics.py:122)"""
    return np.linalg.norm(pose_error(q_gt, t_gt, q_est, t_est)[1], axis=-1)


def abs_rot_err(q_gt, t_gt, q_est, t_est):
    """msg.abs_rot_err = 2 acos |q_w| (diagnostics.py:114,123)"""
    qw = np.abs(pose_error(q_gt, t_gt, q_est, t_est)[0][..., 0]).clip(max=1.0)
    return 2.0 * np.arccos(qw)


def relative_dist_err(abs_err, gt_distance):
    """inf when no distance has been travelled (diagnostics.py:124)"""
    gt_distance = np.asarray(gt_distance, float)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(gt_distance == 0, np.inf, np.asarray(abs_err, float) / gt_distance)


def ate(states_est, states_ref):
    """sqrt(mean_k |t_est - t_ref|^2), no alignment (gauge fixed by the X0 priors) and the max
    rotation error 2 acos|q_w| -- the accuracy metric of BASELINE.json."""
    e, r = np.asarray(states_est, float), np.asarray(states_ref, float)
    d = e[:, 4:7] - r[:, 4:7]
    w = np.abs(np.sum(e[:, :4] * r[:, :4], axis=1)).clip(max=1.0)
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1)))), float(np.max(2 * np.arccos(w)))
