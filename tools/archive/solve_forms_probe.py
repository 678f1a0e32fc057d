"""Times the K4 forms on synthetic windows: usage tools/solve_forms_probe.py <keyframes> <windows> <chunks,chunks,...>
(chunks: 0 = the engine's choice, 1 = whole-window sweeps, P >= 2 = partitioned solve with P chunks).
VF_VARIANT=<path to a diagnostic libvilfusion build> times that build instead."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vil_sensor_fusion_amd import _lib
if os.environ.get('VF_VARIANT'): _lib._SO = os.path.abspath(os.environ['VF_VARIANT'])
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Ps = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 4, 8, 12, 16, 24, 32]
seq = synth.make_sequence(0, N + 8)
rec = synth.between_records(seq)
st = None
for P in Ps:
    eng = Engine(EngineOpts(windows=B, capacity=N + 8, chunks=P))
    for w in range(B):
        eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(w, seq.btw_a, seq.btw_b, rec)
        eng.set_states(w, 0, seq.gt_states[:1]); eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
        if st is None: eng.predict(w, 1, N - 1); st = eng.get_states(0, 0, N)
        else: eng.set_states(w, 0, st)
        eng.set_range(w, 0, N)
    eng.iterate(2)
    a = [eng.time_stage('solve', 10) for _ in range(3)]
    import time
    eng.sync(); t0 = time.perf_counter()
    for _ in range(5): eng.iterate(5)
    eng.sync(); it = (time.perf_counter() - t0) / 5
    print('N', N, 'B', B, 'P', P, 'solve ms', [round(x, 4) for x in a], 'iterate(5) ms', round(it * 1e3, 3), 'cost', eng.read_lm(0)['cost'], flush=True)
    eng.close()
