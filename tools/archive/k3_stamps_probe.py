"""Phase time stamps (s_memtime, shader clock) of ONE workgroup of K3 in the middle of a full batch.
Needs the diagnostic library of tools/build_stamps.sh.  usage: python tools/k3_stamps_probe.py"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vil_sensor_fusion_amd import _lib
_lib._SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvilfusion_stamps.so')
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
B, N = 1024, 1000
seq = synth.make_sequence(0, N)
eng = Engine(EngineOpts(windows=B, capacity=N))
rec = synth.between_records(seq)
for w in range(B):
    eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
    eng.set_between(w, seq.btw_a, seq.btw_b, rec)
    eng.set_states(w, 0, seq.gt_states[:1]); eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
    eng.set_range(w, 0, 1); eng.predict(w, 1, N - 1); eng.set_range(w, 0, N)
eng.linearize(0); eng.decide(init=True); eng.sync()
for rep in range(3):
    eng.assemble(); eng.sync()
    st = (C.c_ulonglong * 8)()
    _lib.lib().vf_debug_k3_stamps(st)
    t = [int(x) for x in st]
    names = ['start -> loads issued', 'loads issued -> own data in LDS', 'barrier', 'MFMA + epilogue (wave 0), stores issued', 'stores acknowledged']
    print(' | '.join(f'{n}: {t[i + 1] - t[i]}' for i, n in enumerate(names)), '| total', t[5] - t[0], 'cycles')
    lp = (C.c_ulonglong * 8)()
    _lib.lib().vf_debug_k3_loop(lp)
    print('   wave 0 loop (3 factors, 2 keyframes):', dict(zip(['top', 'operands LDS->regs', 'Ji^T Ji MFMAs', 'diagonal epilogue + stores', 'off-diagonal part'], [int(x) for x in lp][:5])))
print('assemble ms', eng.time_stage('assemble', 5))
