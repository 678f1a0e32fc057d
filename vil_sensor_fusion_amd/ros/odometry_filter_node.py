#!/usr/bin/env python3
"""Drop-in for the `odometry_filter` node (gtsam_fusion/src/degerate_odometry_filter.cpp:13-51): time-synchronised
(~laser_odom_input nav_msgs/Odometry, ~laser_opt_status loam/OptStatus) pairs, queue 10; the odometry message is
republished on ~laser_odom_output unless the float32 log-det of the rotation or translation block of the 6x6 Hessian is
below ~filter/rot_degen_threshold / ~filter/trans_degen_threshold (launch/fusion.launch:63-64 remaps the topics)."""
from __future__ import annotations

from ..degeneracy_gate import DegeneracyGate


class FilterNode:
    def __init__(self, rospy, message_filters, odometry_cls, opt_status_cls, gate=None):
        self.rospy = rospy
        self.gate = gate or DegeneracyGate(rospy.get_param("~filter/rot_degen_threshold"),
                                           rospy.get_param("~filter/trans_degen_threshold"))
        self.pub = rospy.Publisher("~laser_odom_output", odometry_cls, queue_size=1)
        self.odom_sub = message_filters.Subscriber("~laser_odom_input", odometry_cls, queue_size=1)
        self.status_sub = message_filters.Subscriber("~laser_opt_status", opt_status_cls, queue_size=1)
        self.sync = message_filters.TimeSynchronizer([self.odom_sub, self.status_sub], 10)
        self.sync.registerCallback(self.callback)

    def callback(self, odom, status):
        if self.gate(list(status.hessian)):
            self.pub.publish(odom)
        else:
            self.rospy.loginfo("Degeneracy! odometry at %s dropped" % str(odom.header.stamp))


def main():
    import message_filters
    import rospy
    from loam.msg import OptStatus          # the LOAM fork's message (gtsam_fusion/README.md:19-25)
    from nav_msgs.msg import Odometry
    rospy.init_node("odometry_filter")
    FilterNode(rospy, message_filters, Odometry, OptStatus)
    rospy.spin()


if __name__ == "__main__":
    main()
