"""The committed fixtures of SURVEY.md 8(c) (ii) and (iv) (tests/golden/make_oracle_golden.py): the CPU oracle and the
host-side sensor-manager logic must keep reproducing them (CPU part); K0 and the real GraphManager must match them on the
device (GPU part).  They freeze the restatement; nothing here pins it to GTSAM."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _pim():
    return np.load(os.path.join(GOLD, "pim_testtest.npz"))


def test_oracle_reproduces_the_testtest_recipe(oracle):
    g = _pim()
    i = np.arange(10.0)     # the recipe, restated from gtsam_fusion/test/TestTest.cpp:21-28
    np.testing.assert_array_equal(g["steps"], np.column_stack([np.full(10, 0.01), 0.01 * i, 0.02 * i, 0.03 * i + 9.81,
                                                               0.004 * i, 0.005 * i, 0.006 * i]))
    c = g["covariances"]
    prm = oracle.make_imu_params(*c, gravity=(0, 0, -9.81))
    p = oracle.pim_new(np.zeros(6))
    for s in g["steps"]:
        oracle.pim_integrate(p, prm, s[1:4], s[4:7], s[0])
    rec = oracle.pim_to_record(p)
    np.testing.assert_allclose(rec, g["record"], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(oracle.pim_fields(p)["cov"], g["cov"], rtol=1e-13, atol=1e-300)
    # internal consistency of the fixture: R^T R = information = cov^-1, deltaTij = 0.1 s, first-order sanity of the mean
    np.testing.assert_allclose(g["information"] @ g["cov"], np.eye(15), atol=1e-8)
    assert abs(g["record"][0] - 0.1) < 1e-15
    assert abs(g["record"][9] - 9.81 * 0.1) < 0.02          # delta v_z ~ g * t (plus 0.03 i ramps and rotation)


def test_twin_reproduces_the_testtest_mean():
    from oracle import twin
    g = _pim()
    T, d = twin.preintegrate(g["steps"], np.zeros(6))
    np.testing.assert_allclose(g["record"][0], T, rtol=1e-14)
    np.testing.assert_allclose(g["record"][1:10], d, rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(g["record"][16:70].reshape(9, 6), twin.bias_jacobian_fd(g["steps"], np.zeros(6)), rtol=2e-6, atol=1e-9)


def test_sensor_managers_reproduce_the_integration_timeline():
    from tests.golden.make_oracle_golden import _Recorder
    from tests.test_sensor_manager import _integration_timeline
    from vil_sensor_fusion_amd.sensor_manager import SensorManager
    gold = json.load(open(os.path.join(GOLD, "integration_timeline.json")))
    for name, skip in (("max_time_skip_none", 1e9), ("max_time_skip_carla_0p1", 0.1)):
        gm = _Recorder()
        kw = dict(optimize_after_odom=False, covariance_linear=0.1, covariance_angular=0.01, max_time_skip=skip)
        _integration_timeline(gm, SensorManager(gm, **kw), SensorManager(gm, **kw))
        assert gm.nodes == gold[name]["nodes"]
        assert gm.between == gold[name]["between"]
        assert 3 + len(gm.between) == gold[name]["graph_size_before_solve"]
    assert gold["max_time_skip_none"]["graph_size_before_solve"] == 5 and gold["max_time_skip_none"]["imu_factors_queued"] == 4


@pytest.mark.gpu
def test_k0_matches_the_testtest_fixture():
    from vil_sensor_fusion_amd import Engine, EngineOpts
    g = _pim()
    c = g["covariances"]
    cov = dict(acc=c[0], gyro=c[1], integration=c[2], bias_acc=c[3], bias_omega=c[4], bias_acc_omega_int=c[5])
    eng = Engine(EngineOpts(windows=1, capacity=64))
    eng.preintegrate(0, 1, np.array([0, 10]), g["steps"], np.zeros(6), cov)
    rec = eng.get_imu(0, 1, 1)[0]
    np.testing.assert_allclose(rec[:70], g["record"][:70], rtol=1e-12, atol=1e-15)
    # R by another route (reverse Cholesky + triangular inverse vs inverse + LLT): agree to cond(cov) * eps
    np.testing.assert_allclose(rec[70:], g["record"][70:], rtol=1e-9, atol=1e-9 * np.abs(g["record"][70:]).max())
    R = np.zeros((15, 15))
    R[np.triu_indices(15)] = rec[70:]
    np.testing.assert_allclose(R.T @ R @ g["cov"], np.eye(15), atol=1e-7)
    eng.close()


@pytest.mark.gpu
def test_graph_manager_reproduces_the_integration_timeline():
    from tests.test_sensor_manager import _integration_timeline
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    from vil_sensor_fusion_amd.sensor_manager import SensorManager
    gold = json.load(open(os.path.join(GOLD, "integration_timeline.json")))["max_time_skip_none"]
    gm = GraphManager(capacity=64)
    kw = dict(optimize_after_odom=False, covariance_linear=0.1, covariance_angular=0.01, max_time_skip=1e9)
    _integration_timeline(gm, SensorManager(gm, **kw), SensorManager(gm, **kw),
                          imu=lambda t: gm.addIMUMeasurement(t, [0, 0, 9.81], [0, 0, 0]))
    assert gm.graphSize() == gold["graph_size_before_solve"] and gm.imuQueueSize() == gold["imu_factors_queued"]
    assert list(gm.getMostRecentPoseTime()) == gold["most_recent_pose_time"]
    gm.solve()
    assert gm.graphSize() == 0 and gm.imuQueueSize() == 0          # UnitTests.cpp:385
    # Each factor starts at the previous node's time (the first one at the first IMU stamp, 0.1 s, IMUManager.cpp:76-79) and
    # ends at the LAST IMU sample that had arrived when its node was reserved: the sample that would be interpolated to the
    # node time (IMUManager.cpp:57-66) is not in the buffer yet (IMU every 0.05 s, nodes at 0.47 / 0.87 / 1.07 / 1.27).
    t_nodes = [0.1] + [t for _, t in gold["nodes"]]
    for k in range(1, 5):
        last_sample = np.floor(t_nodes[k] / 0.05 + 1e-9) * 0.05
        np.testing.assert_allclose(gm.imuFactor(k)[0], last_sample - t_nodes[k - 1], rtol=1e-9)
    gm.close()
