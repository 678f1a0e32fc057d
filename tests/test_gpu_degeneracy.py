"""-m gpu: K6 against the reference's own outputs (golden vectors) and, at larger sizes, against
the numpy oracle.  Tolerances: float64 relative 1e-8 on well-conditioned inputs (Jacobi vs
LAPACK), looser where the reference itself is ill-conditioned (see test_oracle_degeneracy.py);
float32: relative 2e-3 on the well-conditioned family only (fp32 tolerance sweep of config 4)."""
import os

import numpy as np
import pytest

from oracle import degeneracy_oracle as dor

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "degeneracy_golden.npz"))
RTOL = {"well": 1e-8, "illcond": 1e-4, "tunnel": 1e-5}
UNSTABLE_WHEN_ILLCOND = {"d_opt_ratio", "a_opt_ratio", "e_opt_ratio", "max_eigen_ratio", "jensen_bregman",
                         "kullback_leibler", "norm_frobenius_ratio", "norm_nuclear_ratio", "norm_1_ratio",
                         "norm_2_ratio"}


@pytest.mark.parametrize("kind", ["well", "illcond", "tunnel"])
@pytest.mark.parametrize("sub", ["all", "trans", "rot"])
def test_k6_matches_reference_golden(kind, sub):
    from vil_sensor_fusion_amd import degeneracy as dg
    mats, pose = GOLD[f"{kind}_mats"], GOLD[f"{kind}_pose"]
    ref = GOLD[f"{kind}_{sub}"]
    worst = 0.0
    for j, name in enumerate(dg.METRICS):
        if name in UNSTABLE_WHEN_ILLCOND and (kind == "illcond" or (kind == "tunnel" and sub != "rot")):
            continue      # kappa^2 >> 1/eps: the reference's own value is rounding noise
        y = dg.apply_degen_function(mats, pose, sub, name)
        assert y[0] == 0.0
        if name == "condition_number" and kind == "illcond":
            np.testing.assert_allclose(y, ref[j], rtol=5e-3)   # kappa * eps
            continue
        if name == "correlation_matrix_distance":
            np.testing.assert_allclose(y, ref[j], atol=1e-15)
            continue
        scale = np.abs(ref[j]).max()
        atol = RTOL[kind] * scale * 1e-3
        if name == "e_opt":
            s = dor.SUBSETS[sub]
            atol = 1e-9 * np.abs(mats[s, s]).max()
        np.testing.assert_allclose(y, ref[j], rtol=RTOL[kind], atol=atol, err_msg=f"{kind}/{sub}/{name}")
        worst = max(worst, np.max(np.abs(y - ref[j]) / (np.abs(ref[j]) + atol)))
    print(f"{kind}/{sub}: worst relative deviation {worst:.2e}")


def test_k6_large_batch_vs_oracle():
    from vil_sensor_fusion_amd import degeneracy as dg
    rng = np.random.default_rng(11)
    T = 20000
    A = rng.normal(size=(T, 6, 6))
    mats = A @ A.transpose(0, 2, 1) + 0.5 * np.eye(6)
    pose = rng.normal(size=(T, 6))
    m3, p3 = mats.transpose(1, 2, 0), pose.T[:, None, :]
    for name in ["d_opt", "e_opt", "max_eigen_ratio", "kullback_leibler", "norm_nuclear_ratio", "condition_number"]:
        for sub in ["all", "rot"]:
            y = dg.apply_degen_function(m3, p3, sub, name)
            ms, ps = dor.subset(mats, pose, sub)
            np.testing.assert_allclose(y, dor.evaluate(name, ms, ps), rtol=1e-7, atol=1e-9, err_msg=f"{name}/{sub}")


def test_k6_fp32_tolerance_sweep():
    from vil_sensor_fusion_amd import degeneracy as dg
    mats, pose = GOLD["well_mats"], GOLD["well_pose"]
    ref = GOLD["well_all"]
    rows = {}
    for name in ["d_opt", "a_opt", "e_opt", "max_eigen", "norm_frobenius", "norm_nuclear", "norm_1", "norm_2"]:
        y32 = dg.apply_degen_function(mats, pose, "all", name, dtype=np.float32)
        j = dg.METRICS.index(name)
        rel = np.abs(y32[1:] - ref[j, 1:]) / np.abs(ref[j]).max()
        rows[name] = rel.max()
        assert rel.max() < 2e-3, name
    print("fp32 vs fp64 reference, worst error relative to the batch max:", {k: f"{v:.1e}" for k, v in rows.items()})


def test_dopt_filter_matches_reference_rule():
    from vil_sensor_fusion_amd import degeneracy as dg
    H = GOLD["filter_hessians_f32"]
    rot, trans, keep = dg.dopt_filter(H)
    ro, to, ko = dor.dopt_filter_f32(H, 11.5, 28.9)
    np.testing.assert_allclose(rot, ro, rtol=1e-6)
    np.testing.assert_allclose(trans, to, rtol=1e-6)
    clear = (np.abs(ro - 11.5) > 1e-4) & (np.abs(to - 28.9) > 1e-4)
    np.testing.assert_array_equal(keep[clear], ko[clear])
    assert 0 < keep.sum() < keep.size


def test_degeneracy_gate_object():
    """vil_sensor_fusion_amd.degeneracy_gate.DegeneracyGate = the callback of degerate_odometry_filter.cpp:29-47: keep unless the
    float32 log det of the rotation (3,3) or translation (0,0) block is below its threshold."""
    from vil_sensor_fusion_amd.degeneracy_gate import DegeneracyGate
    gate = DegeneracyGate(11.5, 28.9)
    good = np.diag([2e4, 3e4, 2.5e4, 60.0, 70.0, 50.0]).astype(np.float32)       # log det: trans 30.3, rot 12.3
    bad_t = good.copy(); bad_t[0, 0] = 2.0                                        # along-track information gone: trans 21.1
    bad_r = good.copy(); bad_r[5, 5] = 5.0                                        # rot 9.95
    keep, rot, trans = gate.evaluate(np.stack([good, bad_t, bad_r]).reshape(3, 36))
    assert keep.tolist() == [True, False, False]
    np.testing.assert_allclose(trans[0], np.log(2e4 * 3e4 * 2.5e4), rtol=1e-5)
    np.testing.assert_allclose(rot[2], np.log(60.0 * 70.0 * 5.0), rtol=1e-5)
    assert gate(good.reshape(36)) and not gate(bad_t.reshape(36)) and gate.dropped == 1


def test_spectrum_is_the_three_metrics_of_one_eigen_solve():
    """vf_degeneracy_spectrum_batch: e_opt, max_eigen, condition_number in one launch, bit for bit the per-metric calls"""
    from vil_sensor_fusion_amd import degeneracy as dg
    for kind in ("well", "illcond", "tunnel"):
        mats = GOLD[f"{kind}_mats"]
        for sub in ("all", "trans", "rot"):
            for dt in (np.float64, np.float32):
                got = dg.spectrum(mats, sub, dtype=dt)
                for name in ("e_opt", "max_eigen", "condition_number"):
                    np.testing.assert_array_equal(got[name], dg.apply_degen_function(mats, None, sub, name, dtype=dt), err_msg=f"{kind}/{sub}/{name}")
    # a matrix that is not symmetric: condition_number is the SVD's business
    m = np.ascontiguousarray(np.tile(np.triu(np.arange(1.0, 37.0).reshape(6, 6))[:, :, None], (1, 1, 70)))
    got = dg.spectrum(m, "all")
    assert np.isnan(got["condition_number"][1:]).all() and np.isfinite(got["e_opt"]).all()
    ref = dg.apply_degen_function(m, None, "all", "condition_number")
    np.testing.assert_allclose(ref[1:], -np.linalg.cond(m[:, :, 1]), rtol=1e-9)


def test_symmetric_eigensolver_on_special_spectra():
    """The tridiagonal QL solver behind e_opt / max_eigen / condition_number / norm_2 / norm_nuclear (round 6) on the spectra that
    break eigen-solvers: the identity and multiples of it (nothing to rotate), the zero matrix, diagonal matrices with repeated
    entries (deflation at every step), rank-one and rank-two matrices (clusters at zero), block-diagonal matrices (an
    off-diagonal that is exactly zero), graded matrices over twelve decades, tight clusters, indefinite matrices, and random
    orthogonal similarity transforms of all of them; float64 against numpy.linalg.eigvalsh, absolute tolerance eps * |A| * 50."""
    from vil_sensor_fusion_amd import degeneracy as dg
    rng = np.random.default_rng(42)
    mats = []

    def add(M):
        mats.append(0.5 * (M + M.T))

    def rot():
        q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
        return q
    add(np.zeros((6, 6)))                                  # index 0 is the series' placeholder (y[0] = 0)
    for s in (1.0, 3.5e7, 2e-9):
        add(s * np.eye(6))
    add(np.zeros((6, 6)))
    add(np.diag([2.0, 2.0, 2.0, 5.0, 5.0, 5.0]))
    add(np.diag([1.0, 1.0, 1.0, 1.0, 1.0, 2.0]))
    add(np.diag([-3.0, 1.0, 4.0, -1.0, 5.0, -9.0]))
    for _ in range(40):
        Q = rot()
        kind = rng.integers(0, 7)
        if kind == 0:
            u = rng.normal(size=6); add(np.outer(u, u))                                  # rank one
        elif kind == 1:
            u, w = rng.normal(size=6), rng.normal(size=6); add(np.outer(u, u) + np.outer(w, w))
        elif kind == 2:
            add(Q @ np.diag(10.0 ** np.linspace(-12, 0, 6)) @ Q.T)                       # graded
        elif kind == 3:
            add(Q @ np.diag(1.0 + 1e-9 * rng.normal(size=6)) @ Q.T)                      # a tight cluster
        elif kind == 4:
            B = np.zeros((6, 6)); a, b = rng.normal(size=(3, 3)), rng.normal(size=(3, 3))
            B[:3, :3], B[3:, 3:] = a @ a.T, b @ b.T; add(B)                              # block diagonal: exact zeros off the blocks
        elif kind == 5:
            add(Q @ np.diag([2.0, 2.0, 2.0, 7.0, 7.0, -1.0]) @ Q.T)                      # repeated, indefinite
        else:
            add(Q @ np.diag(rng.normal(size=6) * 10.0 ** rng.integers(-6, 6)) @ Q.T)
    M = np.ascontiguousarray(np.stack(mats).transpose(1, 2, 0))
    ev = np.linalg.eigvalsh(np.stack(mats))
    scale = np.maximum(np.abs(np.stack(mats)).max(axis=(1, 2)), 1e-300)
    tol = 50 * np.finfo(np.float64).eps * scale
    lo = dg.apply_degen_function(M, None, "all", "e_opt")
    hi = dg.apply_degen_function(M, None, "all", "max_eigen")
    n2 = dg.apply_degen_function(M, None, "all", "norm_2")
    nn = dg.apply_degen_function(M, None, "all", "norm_nuclear")
    assert np.all(np.abs(lo[1:] - ev[1:, 0]) <= tol[1:]), np.abs(lo[1:] - ev[1:, 0]) / tol[1:]
    assert np.all(np.abs(hi[1:] - ev[1:, -1]) <= tol[1:]), np.abs(hi[1:] - ev[1:, -1]) / tol[1:]
    assert np.all(np.abs(n2[1:] - np.abs(ev[1:]).max(axis=1)) <= tol[1:])
    assert np.all(np.abs(nn[1:] - np.abs(ev[1:]).sum(axis=1)) <= 6 * tol[1:])
    # the 3 x 3 subsets run the same code with N = 3
    for sub, sl in (("trans", slice(0, 3)), ("rot", slice(3, 6))):
        ev3 = np.linalg.eigvalsh(np.stack(mats)[:, sl, sl])
        lo3 = dg.apply_degen_function(M, None, sub, "e_opt")
        hi3 = dg.apply_degen_function(M, None, sub, "max_eigen")
        assert np.all(np.abs(lo3[1:] - ev3[1:, 0]) <= tol[1:]) and np.all(np.abs(hi3[1:] - ev3[1:, -1]) <= tol[1:])
    # condition number where it is finite and moderate (|lambda|_max / |lambda|_min); infinite for singular matrices, as numpy's
    finite = np.abs(ev).min(axis=1) > 1e-6 * np.abs(ev).max(axis=1)
    finite[0] = False
    cn = dg.apply_degen_function(M, None, "all", "condition_number")
    ref = -np.abs(ev).max(axis=1) / np.maximum(np.abs(ev).min(axis=1), 1e-300)
    np.testing.assert_allclose(cn[finite], ref[finite], rtol=1e-9)
    print(f"{len(mats) - 1} special matrices: worst |e_opt - eigvalsh| / (eps |A|) = {np.max(np.abs(lo[1:] - ev[1:, 0]) / (np.finfo(np.float64).eps * scale[1:])):.1f}, "
          f"max_eigen {np.max(np.abs(hi[1:] - ev[1:, -1]) / (np.finfo(np.float64).eps * scale[1:])):.1f}")
