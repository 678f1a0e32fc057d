"""-m gpu: SURVEY 8(f)-4 on the device.  The two node adapters of vil_sensor_fusion_amd/ros/ -- the fusion node
(gtsam_fusion_node.cpp:32-98) and the LiDAR degeneracy filter in front of it (degerate_odometry_filter.cpp:23-46) -- wired
as the launch file wires them (launch/fusion.launch:58-73, config/carla/fusion_params.yaml:1-36), with stub ROS plumbing
(tests/ros_stubs.py: topic bus, name resolution / remaps, ROS time, TimeSynchronizer) but the REAL GraphManager and the
REAL DegeneracyGate, i.e. every message ends in libvilfusion.so on the GPU.

A 21 s synthetic Carla drive is replayed as messages at their rates: sensor_msgs/Imu 200 Hz, Image 20 Hz, PointCloud2
10 Hz, Rovio odometry per image, LOAM odometry + loam/OptStatus (6x6 scan-matching Hessian) per cloud, with transport
latencies, LOAM odometry arriving after the VIO solve that already covered its keyframe.  The sequence has a tunnel
stretch (BASELINE configs[3]): the gate must drop exactly those LOAM messages.

Checked: (1) the graph's wiring and the gate's decisions; (2) the factors that reached the GraphManager are the
measurements that were sent; (3) EVERY published ~odometry and TF message against the CPU oracle fed the same factors,
solve for solve (<= 1e-6 m, <= 1e-6 rad); (4) every DiagnosticMessage field of the published estimate against the
ground-truth frame, recomputed independently.
"""
import numpy as np
import pytest

from tests import helpers
from tests import ros_stubs as R
from vil_sensor_fusion_amd import synth

pytestmark = pytest.mark.gpu

PARAMS = {   # gtsam_fusion/config/carla/fusion_params.yaml (both nodes load the same file, launch/fusion.launch:60,71)
    "sensors": {
        # (lidar max_time_skip: the YAML says 0.1 s for a 10 Hz LiDAR; with ROS time arithmetic "stamp difference < 0.1" then
        # holds for about half of the scan pairs.  0.15 keeps every consecutive pair and still rejects a missed scan.)
        "lidar": dict(sensor_topic="/lidar", sensor_type="PointCloud2", odom_topic="/gtsam_fusion_filter/laser_odom_output",
                      optimize_after_odom=False, use_odom_covariance=False, covariance_linear=0.2, covariance_angular=0.2,
                      max_time_skip=0.15),
        "vio": dict(sensor_topic="/cam0/image_mono", sensor_type="Image", odom_topic="/rovio/odometry", optimize_after_odom=True,
                    use_odom_covariance=False, covariance_linear=0.1, covariance_angular=0.1, max_time_skip=0.1)},
    "imu": dict(topic="/imu/fusion", cov_bias_acc=1e-4, cov_bias_omega=1e-6, cov_accel=1e-6, cov_gyro=1e-6, cov_integration=1e-8,
                cov_bias_acc_omega_int=1e-4),
    "tf": dict(static_frame="/rovio_world", odom_frame="/gtsam_odom"),
    "filter": dict(rot_degen_threshold=11.5, trans_degen_threshold=28.9)}
FILTER_REMAPS = {"~laser_odom_input": "/laser_odom_to_init_CORRECTED",          # launch/fusion.launch:63-64
                 "~laser_opt_status": "/laser_odom_optimization_status"}
LATENCY = dict(image=0.004, cloud=0.006, rovio=0.012, loam=0.045, status=0.0455)


def _chain(seq, sensor):
    """absolute odometry of one source: its relative measurements chained from the ground-truth pose of its first keyframe
    (what Rovio / LOAM publish: a drifting pose in their own world frame)"""
    idx = np.nonzero(seq.kf_sensor == sensor)[0]
    Rw, tw = synth.quat_to_rot(seq.gt_states[idx[0], :4]), seq.gt_states[idx[0], 4:7].copy()
    out = {int(idx[0]): (synth.rot_to_quat(Rw), tw.copy())}
    rel = {(int(a), int(b)): (q, t) for a, b, q, t in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t)}
    for a, b in zip(idx[:-1], idx[1:]):
        q, t = rel[(int(a), int(b))]
        tw = tw + Rw @ t
        Rw = Rw @ synth.quat_to_rot(q)
        out[int(b)] = (synth.rot_to_quat(Rw), tw.copy())
    return out


class _Recorder:
    """forwards every call to the real GraphManager and keeps what went through (the test's view of "the same factors")"""

    def __init__(self, gm):
        self.gm, self.nodes, self.between, self.solves = gm, [], [], 0

    def __getattr__(self, k):
        return getattr(self.gm, k)

    def reserveNode(self, t):
        bias = self.gm.getBias()
        key = self.gm.reserveNode(t)
        self.nodes.append((key, t, bias))
        return key

    def addBetweenFactor(self, a, b, pose, cov):
        self.gm.addBetweenFactor(a, b, pose, cov)
        self.between.append((int(a), int(b), np.array(pose[0]), np.array(pose[1]), np.array(cov)))

    def solve(self):
        self.solves += 1
        self.gm.solve()


def test_ros_graph_replay_on_the_device(oracle):
    from vil_sensor_fusion_amd.diagnostics import DiagnosticTrack
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    from vil_sensor_fusion_amd.ros.gtsam_fusion_node import FusionNode
    from vil_sensor_fusion_amd.ros.odometry_filter_node import FilterNode

    n = 640                                                     # 30 keyframes / s: 21.3 s
    seq = synth.make_sequence(seed=21, n_kf=n, tunnel=(0.4, 0.6, 1e-6), keep_raw=True)
    assert seq.kf_time[-1] >= 20.0
    bus = R.Bus()
    msgs = R.ns(Imu="Imu", Image="Image", PointCloud2="PointCloud2", Odometry=R.Msg, TransformStamped=R.Msg)

    # ---- the two nodes, as launch/fusion.launch:58-73 starts them
    rp_filter = R.Rospy(bus, "gtsam_fusion_filter", PARAMS, FILTER_REMAPS)
    filt = FilterNode(rp_filter, R.message_filters_for(rp_filter), "Odometry", "OptStatus")      # real DegeneracyGate (GPU)
    params = dict(PARAMS, solver=dict(lag=1000, iterations=5, rel_tol=0.0, abs_tol=0.0), reference_compat=False)
    rp_node = R.Rospy(bus, "gtsam_fusion_node", params)
    gm = GraphManager(capacity=1192, lag=1000, iterations=5, rel_tol=0.0, abs_tol=0.0)
    gm.setInitialState(seq.gt_states[0])                        # the synthetic vehicle is already moving at t = 0
    rec = _Recorder(gm)
    node = FusionNode(rp_node, R.tf2_ros_for(bus), msgs, graph_manager=rec)
    assert sorted(node.sensor_managers) == ["lidar", "vio"]

    # ---- the feed: (delivery time, order, topic, message)
    rovio, loam = _chain(seq, 0), _chain(seq, 1)
    hess = {int(k): h for k, h in zip(seq.loam_kf, seq.loam_hessians)}
    ev = [(0.0, 0, "/imu/fusion", R.imu_msg(R.Time.from_sec(0.0), seq.imu_acc[0], seq.imu_gyro[0]))]
    for t, a, w in zip(seq.imu_t, seq.imu_acc, seq.imu_gyro):
        ev.append((t, 0, "/imu/fusion", R.imu_msg(R.Time.from_sec(t), a, w)))
    for k in range(n):
        st, t = R.Time.from_sec(seq.kf_time[k]), seq.kf_time[k]
        if seq.kf_sensor[k] == 0:
            ev.append((t + LATENCY["image"], 1, "/cam0/image_mono", R.sensor_msg(st)))
            ev.append((t + LATENCY["rovio"], 2, "/rovio/odometry", R.odometry_msg(st, rovio[k][1], rovio[k][0])))
        else:
            ev.append((t + LATENCY["cloud"], 1, "/lidar", R.sensor_msg(st)))
            ev.append((t + LATENCY["loam"], 2, "/laser_odom_to_init_CORRECTED", R.odometry_msg(st, loam[k][1], loam[k][0])))
            # loam/OptStatus.hessian: 36 floats, row-major, LOAM order [translation, rotation] (degerate_odometry_filter.cpp:30-36)
            ev.append((t + LATENCY["status"], 3, "/laser_odom_optimization_status", R.opt_status_msg(st, hess[k].astype(np.float32).ravel())))
    ev.sort(key=lambda e: (e[0], e[1]))
    t_end = seq.kf_time[-1] + 0.02            # stop before the IMU stream ends (the last keyframes still find their interpolation sample)

    # ---- the oracle, fed the same factors, solve for solve
    g = np.array([0.0, 0.0, -9.81])
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    ostates = np.zeros((n + 1, 16))
    ostates[0] = seq.gt_states[0]
    orecs = np.zeros((n + 1, 190))
    published, worst = [], dict(pos=0.0, rot=0.0)
    state = dict(keys=0, solves_checked=0)

    pending = []        # ~odometry is published from inside vf_solve (the callback runs under the state mutex,
    #                     GraphManager.cpp:117,135-138): the subscriber only queues, the check runs once the solve has returned

    def check_published(m):
        """one message per solve (gtsam_fusion_node.cpp:64-98) against the oracle doing that solve"""
        K = rec.nodes[-1][0]
        for k in range(state["keys"] + 1, K + 1):
            orecs[k] = gm.imuFactor(k)
            ostates[k] = oracle.predict(orecs[k], g, ostates[k - 1])
        state["keys"] = K
        ba = np.array([f[0] for f in rec.between], dtype=np.int32)
        bb = np.array([f[1] for f in rec.between], dtype=np.int32)
        brec = np.zeros((len(rec.between), 28))
        for i, (_, _, q, t, cov) in enumerate(rec.between):
            brec[i, 0:4], brec[i, 4:7], brec[i, 7:28] = q, t, oracle.sqrt_info_upper(cov)
        prob = dict(n=K + 1, states=ostates[:K + 1], imu=orecs[:K + 1], btw_a=ba, btw_b=bb, btw=brec,
                    prior=synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS), gravity=g)
        win = helpers.oracle_window(oracle, prob)
        win.lm(iterations=5)
        ostates[:K + 1] = win.states
        p = np.array([m.pose.pose.position.x, m.pose.pose.position.y, m.pose.pose.position.z])
        q = np.array([m.pose.pose.orientation.w, m.pose.pose.orientation.x, m.pose.pose.orientation.y, m.pose.pose.orientation.z])
        worst["pos"] = max(worst["pos"], float(np.linalg.norm(p - ostates[K, 4:7])))
        worst["rot"] = max(worst["rot"], float(2 * np.arccos(min(1.0, abs(q @ ostates[K, 0:4])))))
        published.append((m.header.stamp.to_sec(), q, p, K))
        state["solves_checked"] += 1

    bus.subs["/gtsam_fusion_node/odometry"].append(pending.append)
    for t, _, topic, m in ev:
        if t > t_end:
            break
        bus.publish(topic, m)
        while pending:
            check_published(pending.pop(0))

    # ---- (1) wiring and gate
    lidar_kf = [k for k in range(1, n) if seq.kf_sensor[k] == 1 and seq.kf_time[k] + LATENCY["status"] <= t_end]
    sent = len(bus.log["/laser_odom_to_init_CORRECTED"])
    passed = bus.log["/gtsam_fusion_filter/laser_odom_output"]
    in_tunnel = {round(seq.kf_time[k], 6) for k in range(n) if seq.kf_sensor[k] == 1 and seq.tunnel[k]}
    passed_stamps = {round(m.header.stamp.to_sec(), 6) for m in passed}
    assert sent == len(lidar_kf) and len(in_tunnel) >= 30
    assert not (passed_stamps & in_tunnel), "the gate let a tunnel scan through"
    assert len(passed) == sent - len(in_tunnel) and filt.gate.dropped == len(in_tunnel) == len(rp_filter.infos)
    assert len(bus.log["/tf"]) == len(bus.log["/gtsam_fusion_node/odometry"]) == rec.solves == state["solves_checked"] >= 400
    assert not node.sensor_managers["vio"].warnings
    o = bus.log["/gtsam_fusion_node/odometry"][-1]
    assert (o.header.frame_id, o.child_frame_id) == ("/rovio_world", "/gtsam_odom")

    # ---- (2) what reached the GraphManager is what was sent
    kf_of_key = {key: int(np.argmin(np.abs(seq.kf_time - t))) for key, t, _ in rec.nodes}
    assert all(abs(seq.kf_time[kf_of_key[key]] - t) < 1e-8 for key, t, _ in rec.nodes)
    rel = {(int(a), int(b)): (q, t, c) for a, b, q, t, c in zip(seq.btw_a, seq.btw_b, seq.btw_q, seq.btw_t, seq.btw_cov)}
    n_lidar_factors = 0
    for a, b, q, t, cov in rec.between:
        qs, ts, c = rel[(kf_of_key[a], kf_of_key[b])]
        assert abs(abs(q @ qs) - 1.0) < 1e-12 and np.abs(t - ts).max() < 1e-9        # poseDiff of the chained odometry
        np.testing.assert_allclose(cov, np.eye(6) * c, rtol=0, atol=0)
        n_lidar_factors += c == synth.LIDAR_COV
        assert not (c == synth.LIDAR_COV and (seq.tunnel[kf_of_key[a]] or seq.tunnel[kf_of_key[b]])), "a gated scan produced a factor"
    assert n_lidar_factors >= 100
    # K0 ran inside the update with the bias estimate of the moment (GraphManager.cpp:59): the record's bhat is getBias() at reserveNode
    for key, _, bias in rec.nodes[::37]:
        np.testing.assert_array_equal(gm.imuFactor(key)[10:16], bias)
    assert np.abs(rec.nodes[-1][2]).max() > 0.0

    # ---- (3) every published message against the oracle
    print(f"ROS replay: {rec.solves} solves, {len(rec.nodes)} keyframes, {len(rec.between)} between factors "
          f"({n_lidar_factors} LiDAR, {len(in_tunnel)} scans gated); published pose vs oracle: worst {worst['pos']:.3e} m, "
          f"{worst['rot']:.3e} rad")
    assert worst["pos"] <= 1e-6 and worst["rot"] <= 1e-6
    K = state["keys"]
    ate, rot = helpers.ate(gm.trajectory(0, K + 1), ostates[:K + 1])
    assert ate <= 1e-6 and rot <= 1e-6
    for tfm, (stamp, q, p, _) in zip(bus.log["/tf"], published):       # TF carries the same pose (gtsam_fusion_node.cpp:85-97)
        assert tfm.header.stamp.to_sec() == stamp and tfm.transform.translation.x == p[0] and tfm.transform.rotation.w == q[0]

    # ---- (4) DiagnosticMessage fields of the published estimate against the ground-truth frame
    track = DiagnosticTrack("fused")
    dist, last_gt, worst_err = 0.0, None, 0.0
    for stamp, q, p, key in published:
        gt = seq.gt_states[kf_of_key[key]]
        d = track.update(stamp, gt[0:4], gt[4:7], q, p)
        if last_gt is not None:
            dist += float(np.linalg.norm(gt[4:7] - last_gt[4:7]))
            Rg = synth.quat_to_rot(gt[0:4])
            np.testing.assert_allclose(d.err_position, Rg.T @ (p - gt[4:7]), atol=1e-12)          # estimate in the gt frame
            assert abs(d.abs_dist_err - np.linalg.norm(p - gt[4:7])) < 1e-12 and abs(d.gt_distance - dist) < 1e-9
            assert abs(d.relative_dist_err - d.abs_dist_err / dist) < 1e-15
            Rl, Re = synth.quat_to_rot(last_gt[0:4]), synth.quat_to_rot(q)
            step_gt = Rl.T @ (gt[4:7] - last_gt[4:7])
            assert abs(d.rel_linear_vel_err * np.linalg.norm(step_gt) - d.abs_linear_vel_err) < 1e-12
            cosang = (np.trace(Rg.T @ Re) - 1) / 2
            assert abs(d.abs_rot_err - np.arccos(np.clip(cosang, -1, 1))) < 1e-6
            worst_err = max(worst_err, d.abs_dist_err)
        last_gt = gt
    print(f"ROS replay: {dist:.1f} m driven, largest abs_dist_err of the published estimate {worst_err:.3f} m")
    assert dist > 200.0 and worst_err < 0.5
    gm.close()


def test_node_builds_its_own_graph_manager_from_parameters():
    """the node constructed the way main() does (no graph_manager argument): the solver/* private parameters reach
    vf_create, solver/initial_state moves the anchor, and one camera keyframe goes through to a published estimate"""
    from vil_sensor_fusion_amd.ros.gtsam_fusion_node import FusionNode
    seq = synth.make_sequence(seed=22, n_kf=12, keep_raw=True)
    bus = R.Bus()
    params = dict(PARAMS, solver=dict(lag=64, capacity=128, iterations=3, rel_tol=0.0, abs_tol=0.0,
                                      initial_state=[float(x) for x in seq.gt_states[0]]))
    rp = R.Rospy(bus, "gtsam_fusion_node", params)
    node = FusionNode(rp, R.tf2_ros_for(bus), R.ns(Imu="Imu", Image="Image", PointCloud2="PointCloud2", Odometry=R.Msg,
                                                   TransformStamped=R.Msg))
    rovio = _chain(seq, 0)
    bus.publish("/imu/fusion", R.imu_msg(R.Time.from_sec(0.0), seq.imu_acc[0], seq.imu_gyro[0]))
    i = 0
    for k in np.nonzero(seq.kf_sensor == 0)[0][:5]:
        while seq.imu_t[i] <= seq.kf_time[k] + 0.01:
            bus.publish("/imu/fusion", R.imu_msg(R.Time.from_sec(seq.imu_t[i]), seq.imu_acc[i], seq.imu_gyro[i]))
            i += 1
        st = R.Time.from_sec(seq.kf_time[k])
        bus.publish("/cam0/image_mono", R.sensor_msg(st))
        bus.publish("/rovio/odometry", R.odometry_msg(st, rovio[int(k)][1], rovio[int(k)][0]))
    out = bus.log["/gtsam_fusion_node/odometry"]
    assert len(out) == 3                      # odometry 0 arms, odometry 1 has no predecessor, 2..4 add a factor and solve
    p = out[-1].pose.pose.position
    gt = seq.gt_states[np.nonzero(seq.kf_sensor == 0)[0][4]]
    assert np.linalg.norm(np.array([p.x, p.y, p.z]) - gt[4:7]) < 0.05
    assert node.graph.lmStats()["accepted"] + node.graph.lmStats()["rejected"] == 9      # 3 solves x 3 trials, no early exit
    node.graph.close()
