"""K6 on 2^22 matrices (the size bench.py's roofline_k6 uses): a few launches of e_opt / condition_number / d_opt in float64 and
float32, for a rocprofv3 --kernel-trace or --pmc pass (tools/profile_k6.sh).  usage: python tools/k6_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vil_sensor_fusion_amd import degeneracy as dg
rng = np.random.default_rng(7)
A = rng.normal(size=(1 << 16, 6, 6))
mats = np.ascontiguousarray((A @ A.transpose(0, 2, 1) + 0.5 * np.eye(6)).transpose(1, 2, 0))
big = np.ascontiguousarray(np.tile(mats, (1, 1, 1 << 6)))
for name in ("e_opt", "condition_number", "d_opt"):
    for dt in (np.float64, np.float32):
        _, ms = dg.apply_degen_function(big, None, "all", name, dtype=dt, reps=5)
        print(f"{name} {np.dtype(dt).name}: {ms:.4f} ms per launch = {ms * 1e6 / (1 << 22):.4f} ns per matrix", flush=True)
