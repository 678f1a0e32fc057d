"""CPU: error definitions of diagnostics.py:103-129 on hand-computable cases."""
import numpy as np

from vil_sensor_fusion_amd import metrics


def test_identity_translation_and_half_turn():
    I = np.array([1.0, 0, 0, 0])
    assert metrics.abs_dist_err(I, [0, 0, 0], I, [0, 0, 0]) == 0
    assert metrics.abs_rot_err(I, [0, 0, 0], I, [0, 0, 0]) == 0
    assert np.isclose(metrics.abs_dist_err(I, [1, 2, 3], I, [2, 4, 5]), 3.0)            # |(1,2,2)|
    half = np.array([0.0, 0, 0, 1.0])                                                   # 180 deg about z
    assert np.isclose(metrics.abs_rot_err(I, [0, 0, 0], half, [0, 0, 0]), np.pi)
    assert np.isclose(metrics.abs_rot_err(half, [0, 0, 0], -half, [0, 0, 0]), 0.0)      # |q_w|: double cover
    # translation error is expressed in the ground-truth frame
    q90 = np.array([np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
    _, t = metrics.pose_error(q90, [0, 0, 0], q90, [1, 0, 0])
    np.testing.assert_allclose(t, [0, -1, 0], atol=1e-15)


def test_relative_error_and_ate():
    assert np.isinf(metrics.relative_dist_err(0.5, 0.0))
    assert metrics.relative_dist_err(0.5, 10.0) == 0.05
    a = np.zeros((4, 16)); a[:, 0] = 1
    b = a.copy(); b[:, 4] = [0.0, 1.0, 1.0, 0.0]
    ate, rot = metrics.ate(a, b)
    assert np.isclose(ate, np.sqrt(0.5)) and rot == 0.0
