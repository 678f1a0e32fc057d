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


class Odometry:
    """The fields of nav_msgs/Odometry the reference reads."""

    def __init__(self, stamp, position, orientation_wxyz, twist_covariance=None):
        self.stamp = float(stamp)
        self.position = np.asarray(position, dtype=np.float64)
        self.orientation = np.asarray(orientation_wxyz, dtype=np.float64)
        self.twist_covariance = None if twist_covariance is None else np.asarray(twist_covariance, dtype=np.float64).reshape(6, 6)


class SensorManager:
    def __init__(self, graph_manager, optimize_after_odom, use_odom_covariance=False,
                 covariance_linear=0.1, covariance_angular=0.1, max_time_skip=0.1, reference_compat=True):
        self.gm = graph_manager
        self.optimize_after_odom = optimize_after_odom
        self.use_odom_covariance = use_odom_covariance
        self.covariance_linear = covariance_linear
        self.covariance_angular = covariance_angular
        self.max_time_skip = max_time_skip
        self.reference_compat = reference_compat
        self.keys_and_times = deque()
        self.last_valid_odom = None
        self.last_valid_key = None
        self.has_received_odometry = False
        self.warnings = []

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
        """addBetweenFactor, failing soft where the device's band is narrower than iSAM2's arbitrary topology
        (GraphManager.cpp:83-88 takes any pair of keys; libvilfusion takes |b - a| <= 3 keyframes and one factor per end
        key -- INTEGRATION.md "Topology limits").  With Carla's max_time_skip = 0.1 s (20 Hz camera + 10 Hz LiDAR keyframes)
        no factor the reference would add is wider; without it (config/san_rafael has no max_time_skip) an odometry gap
        produces one, which is dropped here like a missed odometry message (:41-45 warns and carries on the same way)."""
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
        if self.last_valid_odom is not None and (msg.stamp - self.last_valid_odom.stamp) < self.max_time_skip:
            q, t, tw = self.poseDiff(self.last_valid_odom, msg)
            if self.use_odom_covariance:
                cov_ros = tw.T.copy()       # std::copy into a column-major Matrix66 (:87)
            else:                           # :91-97, filled [lin,lin,lin,ang,ang,ang] as the reference does
                cov_ros = np.diag([self.covariance_linear] * 3 + [self.covariance_angular] * 3)
            added = self._add_between(self.last_valid_key, found[1], (q, t), cov_ros)
            if added and self.optimize_after_odom:
                self.gm.solve()
        self.last_valid_odom = msg
        self.last_valid_key = found[1]
        return added
