"""Stand-ins for rospy / message_filters / tf2_ros / the message classes, enough to run the node adapters of
vil_sensor_fusion_amd/ros/ as a small ROS graph inside one test process (there is no ROS in the image).

What is modelled, after the ROS 1 behaviour the reference's nodes rely on:
  * a Bus = the master's topic table: Publisher.publish delivers synchronously to every Subscriber of the resolved name;
  * private names (`~x` -> /<node>/x) and remaps per node (launch/fusion.launch:63-64);
  * rospy.Time with integer seconds / nanoseconds and to_sec() = secs + nsecs / 1e9 (so stamp arithmetic rounds as it
    does in ROS);
  * message_filters.TimeSynchronizer with the exact-time policy and a bounded queue (degerate_odometry_filter.cpp:23-27).
"""
from __future__ import annotations

import types
from collections import OrderedDict, defaultdict


class Time:
    def __init__(self, secs=0, nsecs=0):
        self.secs, self.nsecs = int(secs), int(nsecs)

    @classmethod
    def from_sec(cls, t):
        secs = int(t)
        return cls(secs, int(round((t - secs) * 1e9)))

    def to_sec(self):
        return float(self.secs) + float(self.nsecs) / 1e9

    def key(self):
        return (self.secs, self.nsecs)

    def __str__(self):
        return f"{self.secs}.{self.nsecs:09d}"


class Msg:
    """attribute bag that grows nested attributes on demand (stands in for any ROS message class)"""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = Msg()
        object.__setattr__(self, k, v)
        return v


def ns(**kw):
    return types.SimpleNamespace(**kw)


class Bus:
    def __init__(self):
        self.subs = defaultdict(list)
        self.log = defaultdict(list)          # every message ever published, per topic

    def publish(self, topic, msg):
        self.log[topic].append(msg)
        for cb in list(self.subs[topic]):
            cb(msg)


class Rospy:
    """one node's view of the graph"""

    def __init__(self, bus, node_name, params, remaps=None):
        self.bus, self.node, self.params, self.remaps = bus, node_name, params, dict(remaps or {})
        self.warned, self.infos, self.errors = [], [], []
        self.Time = Time

    def resolve(self, name):
        name = self.remaps.get(name, name)
        if name.startswith("~"):
            name = "/" + self.node + "/" + name[1:]
        return self.remaps.get(name, name)

    def get_param(self, name, default=None):
        node = self.params
        for part in name.lstrip("~").split("/"):
            if not isinstance(node, dict) or part not in node:
                if default is None:
                    raise KeyError(name)
                return default
            node = node[part]
        return node

    def Subscriber(self, topic, cls, queue_size=None, callback=None):
        topic = self.resolve(topic)
        if callback is not None:
            self.bus.subs[topic].append(callback)
        return ns(topic=topic)

    def Publisher(self, topic, cls, queue_size=None):
        topic = self.resolve(topic)
        return ns(topic=topic, publish=lambda m: self.bus.publish(topic, m))

    def logwarn(self, m):
        self.warned.append(m)

    def loginfo(self, m):
        self.infos.append(m)

    def logerr(self, m):
        self.errors.append(m)


def message_filters_for(rospy):
    """message_filters bound to one node's name resolution"""

    class Subscriber:
        def __init__(self, topic, cls, queue_size=None):
            self.cbs = []
            rospy.Subscriber(topic, cls, queue_size=queue_size, callback=lambda m: [cb(m) for cb in self.cbs])

        def registerCallback(self, cb):
            self.cbs.append(cb)

    class TimeSynchronizer:
        """exact-time policy: a callback fires when every input holds a message with the same header.stamp; each input
        keeps at most `queue` stamps (older ones fall out, as in message_filters::TimeSynchronizer)"""

        def __init__(self, subs, queue):
            self.queues = [OrderedDict() for _ in subs]
            self.queue, self.cb = queue, None
            for i, s in enumerate(subs):
                s.registerCallback(lambda m, i=i: self.add(i, m))

        def registerCallback(self, cb):
            self.cb = cb

        def add(self, i, m):
            q = self.queues[i]
            k = m.header.stamp.key()
            q[k] = m
            while len(q) > self.queue:
                q.popitem(last=False)
            if all(k in qq for qq in self.queues):
                msgs = [qq.pop(k) for qq in self.queues]
                for qq in self.queues:                       # everything older than a matched set is dropped
                    for old in [s for s in qq if s < k]:
                        del qq[old]
                self.cb(*msgs)

    return ns(Subscriber=Subscriber, TimeSynchronizer=TimeSynchronizer)


def tf2_ros_for(bus):
    return ns(TransformBroadcaster=lambda: ns(sendTransform=lambda t: bus.publish("/tf", t)))


def header(stamp):
    return ns(stamp=stamp, frame_id="")


def imu_msg(stamp, acc, gyro):
    return ns(header=header(stamp), linear_acceleration=ns(x=acc[0], y=acc[1], z=acc[2]),
              angular_velocity=ns(x=gyro[0], y=gyro[1], z=gyro[2]))


def sensor_msg(stamp):
    """sensor_msgs/Image or PointCloud2: the node reads the header only (SensorManagerRos.h:91-103)"""
    return ns(header=header(stamp))


def odometry_msg(stamp, position, q_wxyz, twist_covariance=None):
    return ns(header=header(stamp),
              pose=ns(pose=ns(position=ns(x=position[0], y=position[1], z=position[2]),
                              orientation=ns(w=q_wxyz[0], x=q_wxyz[1], y=q_wxyz[2], z=q_wxyz[3]))),
              twist=ns(covariance=list(twist_covariance) if twist_covariance is not None else [0.0] * 36))


def opt_status_msg(stamp, hessian36):
    """loam/OptStatus: the filter reads header.stamp and the 36-float hessian (degerate_odometry_filter.cpp:29-31)"""
    return ns(header=header(stamp), hessian=[float(x) for x in hessian36])
