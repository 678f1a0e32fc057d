"""ATE between the GPU's headline path and the CPU oracle over MANY sequences (bench.py's `accuracy` is window 0 only):
260 windows (one-wave K4 form) x 1000 poses, `S` sampled windows with sequences of their own, a converged start, 12
marginalised warm updates with 5 LM trials.  usage (GPU box): python tools/accuracy_sweep.py [S] [init_iterations]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from tests import helpers
import tests.test_gpu_headline_path as T
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
oracle.build()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
INIT = int(sys.argv[2]) if len(sys.argv) > 2 else 200
UPD, K = 12, 5
sampled = tuple(range(0, 260, 260 // S))[:S]
eng, probs = T._bench_like_engine(oracle, 260, sampled, UPD)
eng.iterate(INIT)
t0 = time.time()
refs = {w: helpers.FixedLagOracle(oracle, probs[w], T.N, K, init_iterations=INIT) for w in sampled}
print(f"oracle initial solves: {time.time() - t0:.1f} s", flush=True)
worst = {w: 0.0 for w in sampled}
for s in range(1, UPD + 1):
    eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
    eng.iterate(K)
    for w in sampled:
        a, r = helpers.ate(eng.get_states(w, s, T.N), refs[w].update())
        worst[w] = max(worst[w], a)
    print(f"update {s:2d}: max over windows {max(helpers.ate(eng.get_states(w, s, T.N), refs[w].window_states)[0] for w in sampled):.3e}", flush=True)
final = [helpers.ate(eng.get_states(w, UPD, T.N), refs[w].window_states)[0] for w in sampled]
print("final ATE per window:", " ".join(f"{x:.1e}" for x in final))
print("worst-over-updates per window:", " ".join(f"{worst[w]:.1e}" for w in sampled))
print(f"init {INIT}: final max {max(final):.3e} median {np.median(final):.3e}; worst over all updates {max(worst.values()):.3e}; windows over 1e-6: {sum(x > 1e-6 for x in worst.values())} of {S}")
