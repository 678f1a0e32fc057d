"""Does splitting the batch over G engines (one HIP stream each) help?  K4 is latency-bound per window and leaves HBM
bandwidth idle, K1 / K3 are bandwidth-bound: kernels of different engines can overlap.  usage: two_stream_probe.py <groups>"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
G = int(sys.argv[1]); B = 1024 // G; N = 1000; STEPS = 6
seqs = [synth.make_sequence(seed=s, n_kf=N + STEPS + 4) for s in range(8)]
recs = [synth.between_records(s) for s in seqs]
engs = []
for g in range(G):
    eng = Engine(EngineOpts(windows=B, capacity=N + STEPS + 4, chunks=1))
    for w in range(B):
        q = (g * B + w) % 8
        seq = seqs[q]
        eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(w, seq.btw_a, seq.btw_b, recs[q])
        eng.set_states(w, 0, seq.gt_states[:1]); eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
    eng.predict(-1, 1, N - 1)
    for w in range(B): eng.set_range(w, 0, N)
    eng.iterate(5); eng.sync()
    engs.append(eng)
def step():
    for e in engs: e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
    for e in engs: e.iterate(5)
for _ in range(2): step()
for e in engs: e.sync()
t0 = time.perf_counter()
for _ in range(STEPS - 2): step()
for e in engs: e.sync()
dt = (time.perf_counter() - t0) / (STEPS - 2)
print('groups', G, 'windows each', B, 'ms per step', round(dt * 1e3, 2), 'keyframes/s', round(1024 / dt))
