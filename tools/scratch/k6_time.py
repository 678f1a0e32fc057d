import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vil_sensor_fusion_amd import degeneracy as dg
rng = np.random.default_rng(7)
A = rng.normal(size=(1 << 16, 6, 6))
mats = np.ascontiguousarray((A @ A.transpose(0, 2, 1) + 0.5 * np.eye(6)).transpose(1, 2, 0))
T2 = 1 << 22
big = np.ascontiguousarray(np.tile(mats, (1, 1, T2 >> 16)))
for name in ("d_opt", "e_opt", "max_eigen", "condition_number", "norm_2", "norm_nuclear", "e_opt_ratio"):
    for dt, tag, nb in ((np.float64, "f64", 296), (np.float32, "f32", 148)):
        _, ms = dg.apply_degen_function(big, None, "all", name, dtype=dt, reps=5)
        print(f"{name:18s} {tag}: {ms:7.3f} ms  {ms * 1e6 / T2:.4f} ns/matrix  {T2 * nb / (ms * 1e-3) / 1e9:7.0f} GB/s  frac {T2 * nb / (ms * 1e-3) / 1e9 / 8000:.3f}")

for dt, tag, nb in ((np.float64, "f64", 296 + 16), (np.float32, "f32", 148 + 8)):
    _, ms = dg.spectrum(big, "all", dtype=dt, reps=5)
    print(f"spectrum (3 outputs)  {tag}: {ms:7.3f} ms  {ms * 1e6 / T2:.4f} ns/matrix  {T2 * nb / (ms * 1e-3) / 1e9:7.0f} GB/s  frac {T2 * nb / (ms * 1e-3) / 1e9 / 8000:.3f}")
