#!/usr/bin/env python3
"""Drop-in for gtsam_fusion/python/diagnostics.py: one thread per entry of ~diagnostics (name, gt, est, ref, rate),
each publishing ~<name> gtsam_fusion/DiagnosticMessage computed by vil_sensor_fusion_amd.diagnostics.DiagnosticTrack from
the two frames' poses in the stationary frame `ref` (TF)."""
from __future__ import annotations

import threading

from ..diagnostics import DiagnosticTrack


def fill_message(msg, d, stamp):
    """DiagnosticTrack output -> gtsam_fusion/DiagnosticMessage (msg/DiagnosticMessage.msg:1-14)"""
    msg.header.stamp = stamp
    for k in ("name", "gt_distance", "abs_dist_err", "abs_rot_err", "relative_dist_err", "abs_linear_vel_err",
              "abs_rot_vel_err", "rel_linear_vel_err", "rel_rot_vel_err"):
        setattr(msg, k, getattr(d, k))
    msg.err.position.x, msg.err.position.y, msg.err.position.z = (float(x) for x in d.err_position)
    (msg.err.orientation.w, msg.err.orientation.x, msg.err.orientation.y,
     msg.err.orientation.z) = (float(x) for x in d.err_orientation)
    return msg


class DiagnosticNode:
    def __init__(self, rospy, tf_listener, message_cls):
        self.rospy, self.tf, self.message_cls = rospy, tf_listener, message_cls
        self.threads = []
        for param in rospy.get_param("~diagnostics"):
            th = threading.Thread(target=self.transform_loop, kwargs=param, daemon=True)
            th.start()
            self.threads.append(th)

    def pose(self, frame, ref, time):
        """pose of `frame` in `ref` at `time`: (q_wxyz, t)"""
        t, q = self.tf.lookupTransform(ref, frame, time)          # tf quaternions are (x, y, z, w)
        return [q[3], q[0], q[1], q[2]], list(t)

    def transform_loop(self, name, gt, est, ref, rate):           # diagnostics.py:33-139
        rospy = self.rospy
        pub = rospy.Publisher("~{}".format(name), self.message_cls, queue_size=1)
        while not rospy.is_shutdown():
            try:
                self.tf.waitForTransform(est, gt, rospy.Time(), rospy.Duration.from_sec(10))
                break
            except Exception:      # noqa: BLE001 -- the reference swallows everything here too (:60)
                rospy.sleep(rospy.Duration.from_sec(0.2))
        track = DiagnosticTrack(name)
        now = self.tf.getLatestCommonTime(gt, est)
        track.update(now.to_sec(), *self.pose(gt, ref, now), *self.pose(est, ref, now))
        last = now
        while not rospy.is_shutdown():
            self.tf.waitForTransform(est, gt, last + rospy.Duration.from_sec((1.0 / rate) * 0.9), rospy.Duration.from_sec(1))
            now = self.tf.getLatestCommonTime(est, gt)
            d = track.update(now.to_sec(), *self.pose(gt, ref, now), *self.pose(est, ref, now))
            if d is not None:
                pub.publish(fill_message(self.message_cls(), d, now))
            last = now


def main():
    import rospy
    from gtsam_fusion.msg import DiagnosticMessage
    from tf import TransformListener
    rospy.init_node("gtsam_fusion_diagnostics")
    DiagnosticNode(rospy, TransformListener(), DiagnosticMessage)
    rospy.spin()


if __name__ == "__main__":
    main()
