"""Seeded synthetic Carla-like workload.

Vehicle on a town loop: speed 8-15 m/s, yaw rate 0.3 sin(2 pi t / 20) rad/s, roll/pitch
+-0.02 rad sinusoids, z = 0.1 sin(2 pi t / 7); IMU 200 Hz
(carla_tools/config/carla_ros_bridge_settings.yaml:12), camera keyframes 20 Hz, LiDAR
keyframes 10 Hz (carla_tools/config/sensors.json:10,100-101); between-factor covariances and
IMU covariances from gtsam_fusion/config/carla/fusion_params.yaml:8-9,17-18,22-27.

Pure numpy, no GPU and no oracle: this module only produces *inputs* (raw IMU steps,
relative-pose measurements, ground truth).  IMU steps are cut exactly as
IMUManager::getFactor does (gtsam_fusion/src/gtsam_fusion/IMUManager.cpp:27-74).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

G = 9.81
IMU_RATE = 200.0
IMU_PHASE = 0.0013          # keeps IMU stamps off the keyframe stamps (no time ties)
CAM_DT, LIDAR_DT, LIDAR_PHASE = 0.05, 0.1, 0.02
VIO_COV, LIDAR_COV = 0.1, 0.2                      # fusion_params.yaml:8-9,17-18
VIO_NOISE = (1e-3, 1e-2)                           # rad, m (1 sigma of the simulated odometry)
LIDAR_NOISE = (5e-4, 5e-3)
IMU_NOISE = 1e-6                                   # sensors.json:108-109
CARLA_IMU_COV = dict(acc=1e-6, gyro=1e-6, integration=1e-8, bias_acc=1e-4, bias_omega=1e-6,
                     bias_acc_omega_int=1e-4)      # fusion_params.yaml:22-27


def _rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    R = np.empty(yaw.shape + (3, 3))
    R[..., 0, 0] = cy * cp; R[..., 0, 1] = cy * sp * sr - sy * cr; R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = sy * cp; R[..., 1, 1] = sy * sp * sr + cy * cr; R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp;     R[..., 2, 1] = cp * sr;                R[..., 2, 2] = cp * cr
    return R


def rot_to_quat(R):
    """(..., 3, 3) -> (..., 4) unit quaternion (w, x, y, z), w >= 0."""
    R = np.asarray(R)
    q = np.empty(R.shape[:-2] + (4,))
    tr = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    q[..., 0] = np.sqrt(np.maximum(1.0 + tr, 1e-300)) / 2
    # the synthetic trajectory never rotates by more than ~100 deg, so w stays large
    q[..., 1] = (R[..., 2, 1] - R[..., 1, 2]) / (4 * q[..., 0])
    q[..., 2] = (R[..., 0, 2] - R[..., 2, 0]) / (4 * q[..., 0])
    q[..., 3] = (R[..., 1, 0] - R[..., 0, 1]) / (4 * q[..., 0])
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def quat_to_rot(q):
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def so3_exp(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    W = np.zeros(w.shape[:-1] + (3, 3))
    W[..., 0, 1] = -w[..., 2]; W[..., 0, 2] = w[..., 1]
    W[..., 1, 0] = w[..., 2];  W[..., 1, 2] = -w[..., 0]
    W[..., 2, 0] = -w[..., 1]; W[..., 2, 1] = w[..., 0]
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    A = np.where(small, 1.0 - th * th / 6, np.sin(ths) / ths)
    B = np.where(small, 0.5 - th * th / 24, (1 - np.cos(ths)) / (ths * ths))
    return np.eye(3) + A * W + B * (W @ W)


class Trajectory:
    """Analytic ground truth; position by fine trapezoid integration of the velocity."""

    def __init__(self, seed: int, duration: float):
        rng = np.random.default_rng([seed, 0xC0FFEE])
        self.ph = rng.uniform(0, 2 * np.pi, size=6)
        self.duration = duration
        fine = 1.0 / 2000.0
        self._tf = np.arange(0.0, duration + 2 * fine, fine)
        v = self.velocity(self._tf)
        self._pf = np.concatenate([np.zeros((1, 3)), np.cumsum(0.5 * (v[1:] + v[:-1]) * fine, axis=0)])

    def yaw(self, t):
        w = 2 * np.pi / 20
        return 0.3 / w * (np.cos(self.ph[0]) - np.cos(w * t + self.ph[0]))

    def yaw_rate(self, t):
        return 0.3 * np.sin(2 * np.pi / 20 * t + self.ph[0])

    def pitch(self, t, d=0):
        w = 2 * np.pi / 5
        return 0.02 * (np.sin(w * t + self.ph[1]) if d == 0 else w * np.cos(w * t + self.ph[1]))

    def roll(self, t, d=0):
        w = 2 * np.pi / 3.7
        return 0.02 * (np.sin(w * t + self.ph[2]) if d == 0 else w * np.cos(w * t + self.ph[2]))

    def speed(self, t, d=0):
        w = 2 * np.pi / 30
        return 11.5 + 3.5 * np.sin(w * t + self.ph[3]) if d == 0 else 3.5 * w * np.cos(w * t + self.ph[3])

    def velocity(self, t):
        psi, s = self.yaw(t), self.speed(t)
        wz = 2 * np.pi / 7
        return np.stack([s * np.cos(psi), s * np.sin(psi), 0.1 * wz * np.cos(wz * t + self.ph[4])], axis=-1)

    def acceleration(self, t):
        psi, dpsi, s, ds = self.yaw(t), self.yaw_rate(t), self.speed(t), self.speed(t, 1)
        wz = 2 * np.pi / 7
        return np.stack([ds * np.cos(psi) - s * dpsi * np.sin(psi), ds * np.sin(psi) + s * dpsi * np.cos(psi),
                         -0.1 * wz * wz * np.sin(wz * t + self.ph[4])], axis=-1)

    def position(self, t):
        t = np.asarray(t, dtype=np.float64)
        return np.stack([np.interp(t, self._tf, self._pf[:, i]) for i in range(3)], axis=-1)

    def rotation(self, t):
        t = np.asarray(t, dtype=np.float64)
        return _rot_zyx(self.yaw(t), self.pitch(t), self.roll(t))

    def body_rate(self, t):
        ph, th = self.roll(t), self.pitch(t)
        dph, dth, dps = self.roll(t, 1), self.pitch(t, 1), self.yaw_rate(t)
        return np.stack([dph - dps * np.sin(th), dth * np.cos(ph) + dps * np.sin(ph) * np.cos(th),
                         -dth * np.sin(ph) + dps * np.cos(ph) * np.cos(th)], axis=-1)

    def specific_force(self, t):
        a = self.acceleration(t) + np.array([0.0, 0.0, G])     # a_world - g, g = (0,0,-9.81) (Z up)
        return np.einsum("...ji,...j->...i", self.rotation(t), a)


def imu_segment(t, acc, gyro, head, start, end):
    """The host half of IMUManager::getFactor (IMUManager.cpp:27-74): which samples are
    dropped / integrated / interpolated.  Returns (steps[n,7] = dt, acc, gyro; new_head)."""
    n = t.shape[0]
    h = head
    pa, pg = np.zeros(3), np.zeros(3)
    while h < n and t[h] <= start:          # :35-40
        pa, pg = acc[h], gyro[h]
        h += 1
    pt = start                              # :44
    steps = []
    while h < n and t[h] < end:             # :46-54
        steps.append(np.concatenate([[t[h] - pt], acc[h], gyro[h]]))
        pt, pa, pg = t[h], acc[h], gyro[h]
        h += 1
    if h < n:                               # :57-66 (the sample stays in the buffer)
        f = (end - pt) / (t[h] - pt)
        steps.append(np.concatenate([[end - pt], f * acc[h] + (1 - f) * pa, f * gyro[h] + (1 - f) * pg]))
    return (np.array(steps) if steps else np.zeros((0, 7))), h


@dataclass
class Sequence:
    seed: int
    kf_time: np.ndarray        # (n,)
    kf_sensor: np.ndarray      # (n,) 0 = camera/VIO, 1 = LiDAR
    gt_states: np.ndarray      # (n,16) ground truth, zero bias
    imu_steps: np.ndarray      # (S,7) dt, acc, gyro for all factors back to back
    imu_off: np.ndarray        # (n+1,) steps of the factor ending at keyframe k: [off[k], off[k+1]); off[0]=off[1]=0
    btw_a: np.ndarray          # (m,)
    btw_b: np.ndarray          # (m,)
    btw_q: np.ndarray          # (m,4) measured relative rotation
    btw_t: np.ndarray          # (m,3)
    btw_cov: np.ndarray        # (m,) isotropic covariance (fusion_params.yaml)
    btw_info: np.ndarray = None        # (m,6) per-dof information multipliers [rot 3, trans 3] (1 = nominal); None = all 1
    tunnel: np.ndarray = None          # (n,) bool: keyframe lies in the LiDAR-degenerate stretch
    loam_hessians: np.ndarray = None   # (n_lidar,6,6) float64: scan-matching Hessian per LiDAR keyframe, LOAM order [trans 3, rot 3]
    loam_kf: np.ndarray = None         # (n_lidar,) keyframe index of each Hessian
    imu_t: np.ndarray = None           # (T,) the raw 200 Hz stream the steps were cut from: stamps,
    imu_acc: np.ndarray = None         # (T,3) specific force,
    imu_gyro: np.ndarray = None        # (T,3) body rate (what a node replaying the sequence as messages publishes)

    @property
    def n(self):
        return self.kf_time.shape[0]


TUNNEL_NOMINAL_EIG = 4.0e4     # scan-matching information per direction: log det of a 3x3 block = 31.8 > 28.9


def make_sequence(seed: int, n_kf: int, vio: bool = True, lidar: bool = True, tunnel=None, keep_raw: bool = False,
                  odom_noise=None) -> Sequence:
    """n_kf keyframes interleaving camera (20 Hz) and LiDAR (10 Hz) stamps; keyframe 0 is the
    anchor (the reference's prior node X(0), GraphManager.cpp:20-35).

    tunnel = (f0, f1, scale): BASELINE configs[3].  For keyframes in the fraction [f0, f1) of the sequence the
    LiDAR odometry is degenerate along the track (body x): the information of its between factors in that
    direction is `scale` x nominal (the measurement noise there grows accordingly), and the per-scan 6x6
    scan-matching Hessians (`loam_hessians`, what loam::OptStatus.hessian carries to the degeneracy filter,
    degerate_odometry_filter.cpp:29-47) lose the same factor in their along-track eigenvalue -- the tunnel stretch
    of the reference's Carla evaluation (DEGEN_TRANS, make_prettier_graphs.py:81-84).

    odom_noise = ((vio_rot, vio_trans), (lidar_rot, lidar_trans)): 1-sigma of the simulated odometry increments instead of
    VIO_NOISE / LIDAR_NOISE (a replay of config/san_rafael, whose factor covariances are 1e-3 ... 1e-7, uses odometry as
    good as those covariances claim)."""
    vio_noise, lidar_noise = odom_noise if odom_noise is not None else (VIO_NOISE, LIDAR_NOISE)
    horizon = n_kf * LIDAR_DT + 1.0      # enough stamps whichever sources are enabled
    cam = np.arange(0, int(horizon / CAM_DT) + 1) * CAM_DT
    lid = np.arange(0, int(horizon / LIDAR_DT) + 1) * LIDAR_DT + LIDAR_PHASE
    times = np.concatenate([cam if vio else [], lid if lidar else []])
    sensor = np.concatenate([np.zeros(cam.size if vio else 0, int), np.ones(lid.size if lidar else 0, int)])
    order = np.argsort(times, kind="stable")
    times, sensor = times[order][:n_kf], sensor[order][:n_kf]
    traj = Trajectory(seed, times[-1] + 1.0)
    rng = np.random.default_rng([seed, 0xBEEF])

    R = traj.rotation(times)
    gt = np.zeros((n_kf, 16))
    gt[:, 0:4] = rot_to_quat(R)
    gt[:, 4:7] = traj.position(times)
    gt[:, 7:10] = traj.velocity(times)

    t_imu = np.arange(0, int((times[-1] + 0.5) * IMU_RATE)) / IMU_RATE + IMU_PHASE
    acc = traj.specific_force(t_imu) + rng.normal(size=(t_imu.size, 3)) * IMU_NOISE
    gyr = traj.body_rate(t_imu) + rng.normal(size=(t_imu.size, 3)) * IMU_NOISE
    steps, off, head = [], [0, 0], 0
    for k in range(1, n_kf):
        s, head = imu_segment(t_imu, acc, gyr, head, times[k - 1], times[k])
        steps.append(s)
        off.append(off[-1] + s.shape[0])
    imu_steps = np.concatenate(steps) if steps else np.zeros((0, 7))

    in_tunnel = np.zeros(n_kf, dtype=bool)
    if tunnel is not None:
        f0, f1, t_scale = tunnel
        in_tunnel[int(f0 * n_kf):int(f1 * n_kf)] = True
    ba, bb, bq, bt, bc, bi = [], [], [], [], [], []
    for sid, (nr, nt), cov in ((0, vio_noise, VIO_COV), (1, lidar_noise, LIDAR_COV)):
        idx = np.nonzero(sensor == sid)[0]
        for a, b in zip(idx[:-1], idx[1:]):
            info = np.ones(6)
            noise_t = rng.normal(size=3) * nt
            if sid == 1 and in_tunnel[b]:
                info[3] = t_scale                        # along-track translation, frame a
                noise_t[0] /= np.sqrt(t_scale) * 1e2     # the scan matcher slides along the tunnel (10 x the nominal noise at 1e-6)
            Rab = R[a].T @ R[b] @ so3_exp(rng.normal(size=3) * nr)
            tab = R[a].T @ (gt[b, 4:7] - gt[a, 4:7]) + noise_t
            ba.append(a); bb.append(b); bq.append(rot_to_quat(Rab)); bt.append(tab); bc.append(cov); bi.append(info)
    o = np.argsort(bb, kind="stable")
    seq = Sequence(seed, times, sensor, gt, imu_steps, np.array(off),
                   np.array(ba, dtype=np.int32)[o], np.array(bb, dtype=np.int32)[o],
                   np.array(bq).reshape(-1, 4)[o], np.array(bt).reshape(-1, 3)[o], np.array(bc)[o])
    if keep_raw:      # the 200 Hz stream itself (a node replay publishes it as messages); off by default: 56 bytes per sample
        seq.imu_t, seq.imu_acc, seq.imu_gyro = t_imu, acc, gyr
    if tunnel is not None:
        seq.btw_info = np.array(bi).reshape(-1, 6)[o]
        seq.tunnel = in_tunnel
        hrng = np.random.default_rng([seed, 0x70AA])
        lk = np.nonzero(sensor == 1)[0]
        Hs = np.zeros((lk.size, 6, 6))
        for i, k in enumerate(lk):
            # J^T J of a scan match: well spread eigenvalues around the nominal, mild rot/trans coupling
            q, _ = np.linalg.qr(hrng.normal(size=(6, 6)) * 0.05 + np.eye(6))
            ev = TUNNEL_NOMINAL_EIG * 10 ** hrng.uniform(-0.15, 0.15, size=6)
            m = (q * ev) @ q.T
            if in_tunnel[k]:
                v = np.zeros(6)                                   # along-track translation (LOAM order: translation first),
                v[:3] = [1.0, 0.1 * hrng.uniform(-1, 1), 0.03 * hrng.uniform(-1, 1)]   # the sensor a few degrees off the tunnel axis
                v /= np.linalg.norm(v)
                mv = m @ v
                m = m - (1 - t_scale) * np.outer(mv, mv) / (v @ mv)   # information along v -> scale x nominal, m stays PSD
            Hs[i] = 0.5 * (m + m.T)
        seq.loam_hessians, seq.loam_kf = Hs, lk
    return seq


def between_records(seq: Sequence) -> np.ndarray:
    """28-double between records with R = chol_upper(cov^-1); the covariances are isotropic
    (SensorManagerRos.cpp:91-97 with use_odom_covariance=false) so R = I / sqrt(cov)."""
    m = seq.btw_a.size
    rec = np.zeros((m, 28))
    rec[:, 0:4] = seq.btw_q
    rec[:, 4:7] = seq.btw_t
    iu = np.triu_indices(6)
    diag_pos = np.nonzero(iu[0] == iu[1])[0]
    rec[:, 7 + diag_pos] = (1.0 / np.sqrt(seq.btw_cov))[:, None]
    if seq.btw_info is not None:
        rec[:, 7 + diag_pos] *= np.sqrt(seq.btw_info)
    return rec


def prior_record(state16, sigmas) -> np.ndarray:
    return np.concatenate([np.asarray(state16, dtype=np.float64), np.asarray(sigmas, dtype=np.float64)])
