#!/usr/bin/env python3
"""Drop-in for the `gtsam_fusion` node (gtsam_fusion/src/gtsam_fusion_node.cpp:17-104): same private parameters
(config/carla/fusion_params.yaml: sensors/<name>/{sensor_topic, sensor_type, odom_topic, optimize_after_odom,
use_odom_covariance, covariance_linear, covariance_angular, max_time_skip}, imu/{topic, cov_*}, tf/{static_frame,
odom_frame}), same subscriptions and queue sizes, same publications (~odometry nav_msgs/Odometry + TF static_frame ->
odom_frame from the optimisation callback), behind Rovio / LOAM exactly like the reference -- the pose-graph arithmetic
runs in libvilfusion.so on the MI355X.

Extra private parameters: solver/lag (fixed-lag window in keyframes, default 1000; 0 = smooth the whole history the way the
reference's unbounded iSAM2 graph does: the engine then grows with the history (vf_engine_grow) and every solve
relinearises all of it, so a solve gets slower as the bag gets longer -- a node meant to run indefinitely keeps a lag),
solver/capacity (default lag + 192, or 4096 initial slots when lag = 0), solver/iterations, solver/rel_tol and
solver/abs_tol (LM termination, default 1e-5 / 1e-5 = gtsam's LevenbergMarquardtParams; 0 / 0 = always `iterations` trials),
solver/device, solver/initial_state (16 doubles q t v bias: the anchor X(0), V(0), B(0) and the means of their priors; default =
the reference's identity / at rest, GraphManager.cpp:20-35), reference_compat (poseDiff quirk), noise_order_compat
(constant-covariance order quirk; both default to true = what the reference does).
NOTE the default solver/lag = 1000 is a deviation: the reference smooths an unbounded iSAM2 graph (= solver/lag 0 here).

Threading: roscpp's ros::spin() runs every callback of the reference node on ONE thread (gtsam_fusion_node.cpp:101).
rospy does not: each subscription delivers on its own receive thread, and ctypes releases the GIL inside the vf_* calls.
SensorManager's bookkeeping (keys_and_times, last_valid_*) has no lock of its own -- the reference relies on that one
spinner thread -- so the node serialises its callbacks with one re-entrant lock (publish() runs inside solve(), on the
thread that holds it).

    rosrun: python3 -m vil_sensor_fusion_amd.ros.gtsam_fusion_node   (with the node's YAML loaded in its namespace,
    launch/fusion.launch:58-73)
"""
from __future__ import annotations

import threading

from ..graph_manager import GraphManager
from ..sensor_manager import Odometry, SensorManager


def odometry_from_msg(m) -> Odometry:
    """nav_msgs/Odometry -> the fields the reference reads (SensorManagerRos.cpp:122-158)"""
    p, q = m.pose.pose.position, m.pose.pose.orientation
    st = m.header.stamp
    ns = int(st.secs) * 10 ** 9 + int(st.nsecs) if hasattr(st, "secs") and hasattr(st, "nsecs") else None
    return Odometry(st.to_sec(), [p.x, p.y, p.z], [q.w, q.x, q.y, q.z], list(m.twist.covariance), stamp_ns=ns)


class FusionNode:
    """Everything of main() that is not ros::init / ros::spin, so that it can be driven by a test with stub messages."""

    def __init__(self, rospy, tf2_ros, msgs, graph_manager=None):
        """msgs: namespace with Imu, Image, PointCloud2, Odometry, TransformStamped message classes"""
        self.rospy, self.msgs = rospy, msgs
        self._lock = threading.RLock()          # one callback at a time, as under ros::spin() (gtsam_fusion_node.cpp:101)
        self._capacity_errors = 0
        P = lambda k, d=None: rospy.get_param("~" + k, d) if d is not None else rospy.get_param("~" + k)
        imu = {k: P("imu/cov_" + n) for k, n in (("acc", "accel"), ("gyro", "gyro"), ("integration", "integration"),
                                                   ("bias_acc", "bias_acc"), ("bias_omega", "bias_omega"),
                                                   ("bias_acc_omega_int", "bias_acc_omega_int"))}   # ImuManagerRos.cpp:20-33
        lag = int(P("solver/lag", 1000))
        capacity = int(P("solver/capacity", lag + 192 if lag > 0 else 4096))
        tol = lambda k: (None if rospy.get_param("~solver/" + k, -1.0) < 0 else float(rospy.get_param("~solver/" + k)))
        self.graph = graph_manager or GraphManager(imu_params=imu, capacity=capacity, lag=lag,
                                                   iterations=int(P("solver/iterations", 5)), device=int(P("solver/device", 0)),
                                                   rel_tol=tol("rel_tol"), abs_tol=tol("abs_tol"),
                                                   # loop closures / wide between factors alive at once (32, the library's default and its limit, unless set lower)
                                                   max_far_factors=(int(P("solver/max_far_factors", 0)) or None))
        x0 = rospy.get_param("~solver/initial_state", [])
        if graph_manager is None and len(x0) == 16:
            self.graph.setInitialState(x0)
        self.subs = [rospy.Subscriber(P("imu/topic"), msgs.Imu, queue_size=100, callback=self._serialised(self.imu_callback))]   # ImuManagerRos.cpp:11
        self.sensor_managers = {}
        compat = bool(P("reference_compat", True))
        noise_compat = bool(P("noise_order_compat", True))      # SURVEY 3.5-2 (sensor_manager.SensorManager.noise_order_compat)
        for name, cfg in sorted(P("sensors").items()):                                   # gtsam_fusion_node.cpp:32-56
            kind = cfg.get("sensor_type")
            if kind not in ("PointCloud2", "Image"):
                rospy.logwarn("Sensor %s has invalid type %s" % (name, kind))            # :52-55
                continue
            use_cov = bool(cfg["use_odom_covariance"])
            if "max_time_skip" not in cfg:
                # the reference reads it with an unchecked getParam into an uninitialised double (SensorManagerRos.h:49,85;
                # config/san_rafael/fusion_params.yaml has none): here "no limit", and a factor that an odometry gap makes
                # wider than the device's band is dropped by SensorManager._add_between
                rospy.logwarn("Sensor %s has no max_time_skip: odometry gaps are not filtered by time" % name)
            sm = SensorManager(self.graph, bool(cfg["optimize_after_odom"]), use_cov,
                               0.0 if use_cov else float(cfg["covariance_linear"]),      # SensorManagerRos.h:50-54
                               0.0 if use_cov else float(cfg["covariance_angular"]),
                               float(cfg.get("max_time_skip", float("inf"))), reference_compat=compat, noise_order_compat=noise_compat)
            self.sensor_managers[name] = sm
            msg_type = msgs.PointCloud2 if kind == "PointCloud2" else msgs.Image
            self.subs.append(rospy.Subscriber(cfg["sensor_topic"], msg_type, queue_size=1,                    # SensorManagerRos.h:59
                                              callback=self._serialised(lambda m, sm=sm: sm.sensorCallback(m.header.stamp.to_sec()))))
            self.subs.append(rospy.Subscriber(cfg["odom_topic"], msgs.Odometry, queue_size=1,                 # :60
                                              callback=self._serialised(lambda m, sm=sm: sm.odometryCallback(odometry_from_msg(m)))))
        self.pub = rospy.Publisher("~odometry", msgs.Odometry, queue_size=1)             # gtsam_fusion_node.cpp:58
        self.broadcaster = tf2_ros.TransformBroadcaster()
        self.static_frame, self.odom_frame = P("tf/static_frame"), P("tf/odom_frame")    # :61-62
        self.graph.addOptimizationCallback(self.publish)                                 # :64

    def _serialised(self, fn):
        """fn under the node's lock.  VF_ERR_CAPACITY from a callback (no keyframe slot left with solver/capacity fixed, or a
        window that cannot hold solver/lag plus the keyframes of one solve) does not kill the subscriber thread: it is
        logged -- the first occurrence and then every 100th, with the running count, so that a persistent failure keeps
        showing -- and the message is dropped.  (A between factor the band cannot hold never gets here:
        SensorManager._add_between handles that case.)  Every other error propagates."""
        from .._lib import VilFusionError

        def call(m):
            with self._lock:
                try:
                    return fn(m)
                except VilFusionError as exc:
                    if exc.code != -6:                      # VF_ERR_CAPACITY
                        raise
                    self._capacity_errors += 1
                    if self._capacity_errors == 1 or self._capacity_errors % 100 == 0:
                        log = getattr(self.rospy, "logerr", None) or self.rospy.logwarn
                        log("gtsam_fusion: %s (capacity error #%d: the message is dropped) -- check ~solver/lag and "
                            "~solver/capacity" % (exc, self._capacity_errors))
                    return None
        return call

    def imu_callback(self, m):                                                           # ImuManagerRos.cpp:38-52
        a, w = m.linear_acceleration, m.angular_velocity
        self.graph.addIMUMeasurement(m.header.stamp.to_sec(), [a.x, a.y, a.z], [w.x, w.y, w.z])

    def publish(self, time, q, p, v, bias):                                              # gtsam_fusion_node.cpp:64-98
        stamp = self.rospy.Time.from_sec(time)
        o = self.msgs.Odometry()
        o.header.stamp, o.header.frame_id, o.child_frame_id = stamp, self.static_frame, self.odom_frame
        o.pose.pose.position.x, o.pose.pose.position.y, o.pose.pose.position.z = (float(x) for x in p)
        (o.pose.pose.orientation.w, o.pose.pose.orientation.x, o.pose.pose.orientation.y,
         o.pose.pose.orientation.z) = (float(x) for x in q)
        o.twist.twist.linear.x, o.twist.twist.linear.y, o.twist.twist.linear.z = (float(x) for x in v)
        self.pub.publish(o)
        t = self.msgs.TransformStamped()
        t.header.stamp, t.header.frame_id, t.child_frame_id = stamp, self.static_frame, self.odom_frame
        t.transform.translation.x, t.transform.translation.y, t.transform.translation.z = (float(x) for x in p)
        (t.transform.rotation.w, t.transform.rotation.x, t.transform.rotation.y,
         t.transform.rotation.z) = (float(x) for x in q)
        self.broadcaster.sendTransform(t)


def main():
    import types

    import rospy
    import tf2_ros
    from geometry_msgs.msg import TransformStamped
    from nav_msgs.msg import Odometry as OdometryMsg
    from sensor_msgs.msg import Image, Imu, PointCloud2
    rospy.init_node("gtsam_fusion")
    FusionNode(rospy, tf2_ros, types.SimpleNamespace(Imu=Imu, Image=Image, PointCloud2=PointCloud2, Odometry=OdometryMsg,
                                                      TransformStamped=TransformStamped))
    rospy.spin()          # rospy delivers each subscription on its own thread: FusionNode serialises them (see the module docstring)


if __name__ == "__main__":
    main()
