"""s_memtime stamps of the forward sweep's step (diagnostic build, tools/build_stamps.sh): split form against the assembling form.
usage (GPU box): python tools/asm_stamps_probe.py B"""
import sys, ctypes as C, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vil_sensor_fusion_amd import _lib
_lib._SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvilfusion_stamps.so')
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
B = int(sys.argv[1]); N = 1000
seq = synth.make_sequence(0, N)
rec = synth.between_records(seq)
for asm, waves in ((0, 1), (1, 1), (1, 2)):
    eng = Engine(EngineOpts(windows=B, capacity=N, chunks=1, sweep_two_sided_max=0, solve_split_min=1, solve_assemble_min=asm, solve_assemble_waves=waves))
    for w in range(B):
        eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(w, seq.btw_a, seq.btw_b, rec)
        eng.set_states(w, 0, seq.gt_states[:1]); eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
    eng.predict(-1, 1, N - 1)
    for w in range(B):
        eng.set_range(w, 0, N)
    eng.linearize(0); eng.decide(init=True); eng.assemble(); eng.sync()
    print('B', B, ('assembling, %d wave(s) per window' % waves) if asm else 'split', 'solve ms', eng.time_stage('solve', 3))
    st = (C.c_ulonglong * 16)()
    _lib.lib().vf_debug_solve_stamps(st)
    names = ['loop top (+ as_advance)', 'panel row LDS load', 'pivot chain (+ pieces)', 'P write + panel store', 'schur mfma + commit', 'write-back',
             'bs: between steps', 'bs prep: rows->LDS (+ panel wait)', 'bs prep: column data + far couplings', 'bs: (prep -> solve)', 'bs solve: register chain + store']
    tot = sum(st[:6])
    for i, nme in enumerate(names):
        print(f'{nme:38s} {st[i]/N:9.1f} ticks/step  {100*st[i]/tot:5.1f}% of the forward sweep')
    eng.close()
