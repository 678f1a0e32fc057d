"""rospy adapters: the gtsam_fusion topic surface over the C ABI.  Importing a node module needs ROS;
the logic they wire up (GraphManager, SensorManager, DiagnosticTrack, DegeneracyGate) is ROS-free and tested without it."""
