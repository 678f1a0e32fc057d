"""-m gpu: a hand-shake that never gets its answer must end as a FAILED SOLVE of that window, not as a wave that spins until
the GPU is reset.  The kernels whose waves wait for each other through LDS cells -- the two-wave
assembling sweep (eliminator <-> assembler) and the partitioned solve's chunk kernel (sweep <-> spike follower) -- bound every
wait (2^22 polls).  tools/variants/libvilfusion_withhold.so is the library built from the same sources with
-DVF_RING_WITHHOLD (by __graft_entry__.build(); test infrastructure, never loaded by the product): in it the spike follower
of chunk 1 of window 0 is told its producer never published, and the eliminator of window 1 that its assembler never
answered.  Each case runs in a child process (the loader is pointed at the variant before its first use)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "tools", "variants", "libvilfusion_withhold.so")

CHILD = r'''
import json, sys
sys.path.insert(0, %(root)r)
from vil_sensor_fusion_amd import _lib
_lib._SO = %(variant)r
import numpy as np
from oracle import oracle
from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
oracle.build()
mode = sys.argv[1]
n, B = 120, 3
opts = dict(chunks=4) if mode == "partitioned" else dict(chunks=1, sweep_two_sided_max=0, solve_assemble_min=1, solve_assemble_waves=2)
eng = Engine(EngineOpts(windows=B, capacity=n, **opts))
probs = [helpers.build_problem(oracle, synth.make_sequence(seed=40 + w, n_kf=n), perturb=0.01) for w in range(B)]
for w in range(B):
    helpers.load_engine(eng, w, probs[w])
eng.iterate(3)
out = {"form": eng.solve_form(), "lm": [eng.read_lm(w) for w in range(B)], "ate": []}
for w in range(B):
    win = helpers.oracle_window(oracle, probs[w])
    win.lm(iterations=3)
    out["ate"].append(helpers.ate(eng.get_states(w, 0, n), win.states)[0])
    out.setdefault("moved", []).append(float(np.abs(eng.get_states(w, 0, n) - probs[w]["states"]).max()))
print("RESULT " + json.dumps(out))
'''


def _run(mode):
    if not os.path.exists(VARIANT):
        pytest.fail(f"{VARIANT} missing: run __graft_entry__.build()")
    res = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "variant": VARIANT}, mode], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.parametrize("mode,victim", [("partitioned", 0), ("two_wave_sweep", 1)])
def test_a_withheld_handshake_is_a_failed_solve_of_that_window_only(mode, victim):
    out = _run(mode)
    print(mode, out)
    assert out["form"] == ("partitioned" if mode == "partitioned" else "assembling")
    for w, lm in enumerate(out["lm"]):
        if w == victim:
            # every trial of the victim window is a failed solve: rejected, its states never move
            assert lm["solve_failures"] == 3 and lm["accepted"] == 0 and out["moved"][w] == 0.0
        else:
            # the neighbours solve as if nothing had happened
            assert lm["solve_failures"] == 0 and lm["accepted"] >= 1 and out["ate"][w] <= 1e-6
