#!/usr/bin/env python3
"""Generates tests/golden/degeneracy_golden.npz by IMPORTING the reference's own Python
(/root/reference/vil_fusion/python/degeneracy_detection_functions.py and the batched caller
apply_degen_function of make_prettier_graphs.py) in this container, with ROS modules stubbed.
Only inputs and outputs are stored; no reference source travels.  Run here (not on the GPU box):

    python tests/golden/make_degeneracy_golden.py
"""
import importlib.util
import math
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/vil_fusion/python"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "degeneracy_golden.npz")

for name in ("rospy", "vil_fusion", "vil_fusion.msg", "nav_msgs", "nav_msgs.msg", "rosbag", "tf",
             "tf.transformations", "matplotlib", "matplotlib.pyplot", "matplotlib.lines", "matplotlib.patches",
             "matplotlib.ticker", "matplotlib.font_manager", "matplotlib.axes"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["vil_fusion.msg"].DegeneracyScore = object
sys.modules["nav_msgs.msg"].Odometry = object


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


funcs = load(os.path.join(REF, "degeneracy_detection_functions.py"), "ref_degen_funcs")

# apply_degen_function is pure numpy; lift just that function out of make_prettier_graphs.py by
# executing its source text in an empty namespace (the module itself needs rosbag/matplotlib).
src = open(os.path.join(REF, "make_prettier_graphs.py")).read()
start = src.index("def apply_degen_function(")
end = src.index("def calc_roc(")
ns = {"np": np}
exec(compile(src[start:end], "apply_degen_function", "exec"), ns)
apply_degen_function = ns["apply_degen_function"]


def spd(rng, n, cond, scale):
    q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    ev = scale * np.logspace(0, -math.log10(cond), n)
    return (q * ev) @ q.T


def batch(rng, T, kind):
    mats = np.zeros((6, 6, T))
    for i in range(T):
        if kind == "well":
            m = spd(rng, 6, 10 ** rng.uniform(0.5, 3), 10 ** rng.uniform(2, 6))
        elif kind == "illcond":
            m = spd(rng, 6, 1e12, 1e6)
        elif kind == "tunnel":      # LOAM-like ICP Hessian, one translational direction ~1e-6 x nominal
            m = spd(rng, 6, 1e2, 1e5)
            v = np.zeros(6); v[:3] = rng.normal(size=3); v /= np.linalg.norm(v)
            mv = m @ v                      # information along v drops to 1e-6 x nominal, m stays PSD
            m = m - (1 - 1e-6) * np.outer(mv, mv) / (v @ mv)
        m = 0.5 * (m + m.T)
        mats[:, :, i] = m
    return mats


names = [f.__name__ for f in funcs.degen_funcs] + ["condition_number", "differential_entropy"]
all_funcs = list(funcs.degen_funcs) + [funcs.condition_number, funcs.differential_entropy]
rng = np.random.default_rng(20260101)
out = {"names": np.array(names)}
warnings.simplefilter("ignore")
for kind, T in (("well", 48), ("illcond", 24), ("tunnel", 24)):
    mats = batch(rng, T, kind)
    pose = rng.normal(size=(6, 1, T)) * np.array([5, 5, 1, 0.1, 0.1, 1.0]).reshape(6, 1, 1)
    out[f"{kind}_mats"] = mats
    out[f"{kind}_pose"] = pose
    for subset in ("all", "trans", "rot"):
        res = np.zeros((len(all_funcs), T))
        for j, f in enumerate(all_funcs):
            y = apply_degen_function(mats, pose, subset, f)
            res[j] = np.real(y)
        out[f"{kind}_{subset}"] = res

# the shipped float32 D-optimality filter (gtsam_fusion/src/degerate_odometry_filter.cpp:29-47)
# has no Python form in the reference; its golden is the float32 restatement of the C++ lambda
# evaluated by numpy in float32 on Hessians scaled to straddle the shipped thresholds 11.5 / 28.9.
H = batch(rng, 64, "well").transpose(2, 0, 1)
H[:, :3, :3] *= 40.0
out["filter_hessians_f32"] = H.astype(np.float32)
np.savez_compressed(OUT, **out)
print("wrote", OUT, {k: v.shape for k, v in out.items() if k != "names"})
print(names)
