"""-m gpu: K0's 15 x 15 covariance recursion (k_preintegrate; replaces PreintegratedCombinedMeasurements::
integrateMeasurement as driven by IMUManager.cpp:42-64) against the INDEPENDENT propagation of oracle/twin.py, whose
sensitivities are central differences of its own one-sample step -- not the closed-form A, B, C blocks that K0 and the C
oracle both restate.  The device record holds R = chol_upper(preintMeasCov^-1); compared as R^T R P_twin = I and as the
covariance itself, on the TestTest.cpp:11-29 recipe, a San-Rafael-parameter sequence and a long Carla one."""
import numpy as np
import pytest

from oracle import twin
from tests.test_oracle_twin import _twin_cov, covariance_cases
from vil_sensor_fusion_amd import Engine, EngineOpts

pytestmark = pytest.mark.gpu


def test_k0_covariance_matches_the_independent_twin(oracle):
    cases = covariance_cases()
    eng = Engine(EngineOpts(windows=1, capacity=64))
    for i, (name, steps, bhat, c) in enumerate(cases):
        eng.preintegrate(0, 1 + i, np.array([0, len(steps)]), steps, bhat, c)
        rec = eng.get_imu(0, 1 + i, 1)[0]
        Pt = _twin_cov(steps, bhat, c)
        T, d = twin.preintegrate(steps, bhat)
        assert abs(rec[0] - T) <= 1e-14 and np.abs(rec[1:10] - d).max() <= 1e-11      # the mean, while we are here
        R = oracle.unpack_upper(rec[70:], 15)
        info = R.T @ R
        resid = np.abs(info @ Pt - np.eye(15)).max()
        P = np.linalg.inv(info)
        sd = np.sqrt(np.diag(Pt))
        err = (np.abs(P - Pt) / np.outer(sd, sd)).max()
        print(f"{name}: |R^T R P_twin - I| {resid:.3e}; correlation-scaled covariance difference {err:.3e}")
        assert resid < 5e-6 and err < 5e-6, name
    eng.close()
