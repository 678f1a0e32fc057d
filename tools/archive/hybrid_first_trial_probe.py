"""Why is the sweep of the FIRST trial under the termination rule slower than the same sweep without the rule?  Stage time of
the solve (all 1 024 windows active) with the rule off / on, the compacted active list on / off.
usage (GPU box): python tools/hybrid_first_trial_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("VF_LIB"):
    from vil_sensor_fusion_amd import _lib
    _lib._SO = os.path.abspath(os.environ["VF_LIB"])
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS

n, B = 1000, 1024
seqs = [synth.make_sequence(seed=80 + i, n_kf=n + 8) for i in range(16)]
for active_list in (1, 0):
    eng = Engine(EngineOpts(windows=B, capacity=n + 8, hybrid_active_list=active_list))
    recs = [synth.between_records(s) for s in seqs]
    for w in range(B):
        s = seqs[w % len(seqs)]
        eng.preintegrate(w, 1, s.imu_off[1:n + 1], s.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        m = s.btw_b < n
        eng.set_between(w, s.btw_a[m], s.btw_b[m], recs[w % len(seqs)][m])
        eng.set_states(w, 0, s.gt_states[0].reshape(1, 16))
        eng.set_prior(w, 0, synth.prior_record(s.gt_states[0], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
    eng.predict(-1, 1, n - 1)
    for w in range(B):
        eng.set_range(w, 0, n)
    eng.iterate(3)
    off = [eng.time_stage("solve", 5) for _ in range(3)]
    eng.set_convergence(1e-5, 1e-5)
    eng.linearize(); eng.decide(init=True)
    on = [eng.time_stage("solve", 5) for _ in range(3)]
    print(f"active list {active_list}: solve stage, rule off {' '.join(f'{x:.3f}' for x in off)} ms; rule on (every window active) {' '.join(f'{x:.3f}' for x in on)} ms", flush=True)
    eng.close()
