"""ROS-free replay of SensorManagerRos (gtsam_fusion/src/gtsam_fusion/SensorManagerRos.cpp:11-158,
include/gtsam_fusion/SensorManagerRos.h:90-103): one instance per odometry source.

Host bookkeeping only (stamp<->key matching, first-odometry gating, max_time_skip, poseDiff,
covariance selection); the factors it produces are evaluated on the GPU by the GraphManager.
`reference_compat=True` reproduces the reference's poseDiff exactly, including its world-frame
rotation delta q2*q1^-1 (SensorManagerRos.cpp:148); False uses the Pose3 between R1^T R2.
"""
from __future__ import annotations

from collections import deque

import numpy as np


def _qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _qinv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]]) / np.dot(q, q)


def _qrot(q, v):
    return _qmul(_qmul(q, np.concatenate([[0.0], v])), _qinv(q))[1:]


def ros_duration_sec(later_ns: int, earlier_ns: int) -> float:
    """(later - earlier).toSec() as roscpp computes it: the difference of two ros::Time is a ros::Duration in INTEGER
    (sec, nsec), and toSec() = sec + 1e-9 * nsec (rostime duration.h).  Matters at the boundary: stamps exactly 0.1 s apart
    give exactly the double 0.1, and `0.1 < max_time_skip` with the YAML's 0.1 is false (SensorManagerRos.cpp:47)."""
    d = int(later_ns) - int(earlier_ns)
    sec, nsec = divmod(d, 10 ** 9)
    return float(sec) + 1e-9 * float(nsec)


class Odometry:
    """The fields of nav_msgs/Odometry the reference reads.  stamp_ns: header.stamp as integer nanoseconds (what ros::Time
    holds); when absent it is the float stamp rounded to the nanosecond (exact for simulated time, not for epoch stamps)."""

    def __init__(self, stamp, position, orientation_wxyz, twist_covariance=None, stamp_ns=None):
        self.stamp = float(stamp)
        self.stamp_ns = int(round(self.stamp * 1e9)) if stamp_ns is None else int(stamp_ns)
        self.position = np.asarray(position, dtype=np.float64)
        self.orientation = np.asarray(orientation_wxyz, dtype=np.float64)
        self.twist_covariance = None if twist_covariance is None else np.asarray(twist_covariance, dtype=np.float64).reshape(6, 6)


class SensorManager:
    def __init__(self, graph_manager, optimize_after_odom, use_odom_covariance=False,
                 covariance_linear=0.1, covariance_angular=0.1, max_time_skip=0.1, reference_compat=True,
                 noise_order_compat=True):
        self.gm = graph_manager
        self.optimize_after_odom = optimize_after_odom
        self.use_odom_covariance = use_odom_covariance
        self.covariance_linear = covariance_linear
        self.covariance_angular = covariance_angular
        self.max_time_skip = max_time_skip
        self.reference_compat = reference_compat
        # the reference fills the constant covariance [lin, lin, lin, ang, ang, ang] (SensorManagerRos.cpp:91-97)
        # and hands it to a factor whose tangent order is [rot, trans]: the rotation gets covariance_linear, the translation
        # covariance_angular.  True reproduces that (benign in config/carla: 0.2 / 0.2, 0.1 / 0.1; not in config/san_rafael:
        # 1e-6 / 1e-7, 1e-3 / 1e-4); False puts each where its name says.
        self.noise_order_compat = noise_order_compat
        # diagnostics kept for the caller, bounded (a node runs for hours: the last `keep` entries of each)
        keep = 4096
        self.skipped = deque(maxlen=keep)    # (previous stamp, stamp) of consecutive odometry messages the max_time_skip test refused
        self.keys_and_times = deque()
        self.last_valid_odom = None
        self.last_valid_key = None
        self.has_received_odometry = False
        self.warnings = deque(maxlen=keep)

    # SensorManagerRos.h:91-103
    def sensorCallback(self, stamp):
        if self.has_received_odometry:
            key = self.gm.reserveNode(float(stamp))
            self.keys_and_times.append((float(stamp), key))
            return key
        return None

    # SensorManagerRos.cpp:122-158
    def poseDiff(self, before: Odometry, after: Odometry):
        dx = after.position - before.position
        dxr = _qrot(_qinv(before.orientation), dx)
        if self.reference_compat:
            qr = _qmul(after.orientation, _qinv(before.orientation))      # :148 (world-frame delta)
        else:
            qr = _qmul(_qinv(before.orientation), after.orientation)
        return qr, dxr, after.twist_covariance

    def _add_between(self, a, b, pose, cov):
        """addBetweenFactor, failing soft where the library cannot take a factor iSAM2 would (GraphManager.cpp:83-88 takes any
        pair of keys).  The library takes any pair too -- a span wider than 3 keyframes or a second factor on an end key
        becomes a "far" factor (vf_engine_set_extra_between) -- but holds at most 32 of those alive per window
        (vf_graph_opts.max_far_factors; INTEGRATION.md "Limits"): one more comes back as VF_ERR_CAPACITY and is dropped here like a missed odometry
        message (:41-45 warns and carries on the same way)."""
        from ._lib import VilFusionError
        try:
            self.gm.addBetweenFactor(a, b, pose, cov)
            return True
        except VilFusionError as exc:
            if exc.code != -6:                  # VF_ERR_CAPACITY: span / second factor on a key
                raise
            self.warnings.append(f"between factor ({a}, {b}) not added: {exc}")
            return False

    # SensorManagerRos.cpp:11-120
    def odometryCallback(self, msg: Odometry):
        if not self.has_received_odometry:
            self.has_received_odometry = True
            return False
        found = None
        while self.keys_and_times and found is None:
            t, key = self.keys_and_times[0]
            if t > msg.stamp:
                break
            self.keys_and_times.popleft()
            if abs(round((t - msg.stamp) * 1e9)) < 1000000:       # :34, 1 ms in ns
                found = (t, key)
        if found is None:
            self.warnings.append(f"odometry at {msg.stamp} has no corresponding key")   # :41-45
            return False
        added = False
        # :47 -- strict "<" on a ros::Duration: a 10 Hz source with exact stamps and the YAML's max_time_skip = 0.1
        # (config/carla/fusion_params.yaml:10) never passes it
        within = self.last_valid_odom is not None and ros_duration_sec(msg.stamp_ns, self.last_valid_odom.stamp_ns) < self.max_time_skip
        if self.last_valid_odom is not None and not within:
            self.skipped.append((self.last_valid_odom.stamp, msg.stamp))
        if within:
            q, t, tw = self.poseDiff(self.last_valid_odom, msg)
            if self.use_odom_covariance:
                cov_ros = tw.T.copy()       # std::copy into a column-major Matrix66 (:87)
            elif self.noise_order_compat:   # :91-97, filled [lin,lin,lin,ang,ang,ang] as the reference does
                cov_ros = np.diag([self.covariance_linear] * 3 + [self.covariance_angular] * 3)
            else:                           # Pose3 tangent order [rot, trans]
                cov_ros = np.diag([self.covariance_angular] * 3 + [self.covariance_linear] * 3)
            added = self._add_between(self.last_valid_key, found[1], (q, t), cov_ros)
            if added and self.optimize_after_odom:
                self.gm.solve()
        self.last_valid_odom = msg
        self.last_valid_key = found[1]
        return added
