import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle
from tests import helpers
from vil_sensor_fusion_amd import Engine, EngineOpts, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seq = synth.make_sequence(0, N)
prob = helpers.build_problem(oracle, seq, perturb=0.01)
eng = Engine(EngineOpts(windows=1, capacity=max(64, N)))
helpers.load_engine(eng, 0, prob)
eng.linearize(0); eng.assemble(); eng.solve()
H, g = eng.read_normal(0, 0, N); P = eng.read_panels(0, 0, N)
A = np.zeros((15 * N, 15 * N), dtype=np.longdouble)
for k in range(N):
    for d in range(min(k, 3) + 1):
        A[15*k:15*k+15, 15*(k-d):15*(k-d)+15] = H[k, d]
        A[15*(k-d):15*(k-d)+15, 15*k:15*k+15] = H[k, d].T
A += np.eye(15 * N) * 1e-5
# longdouble cholesky
n = 15 * N
L = np.zeros_like(A)
for j in range(n):
    s = A[j, j] - L[j, :j] @ L[j, :j]
    L[j, j] = np.sqrt(s)
    if j + 1 < n:
        hi = min(n, j + 61)
        L[j+1:hi, j] = (A[j+1:hi, j] - L[j+1:hi, :j] @ L[j, :j]) / L[j, j]
y = np.linalg.solve(L.astype(np.float64), -g.reshape(-1))
np.set_printoptions(linewidth=220, precision=2)
worst=[]
for k in range(N):
    rows = list(range(15*k, 15*k+15)) + list(range(15*(k+1), 15*(k+1)+15)) + list(range(15*(k+2), 15*(k+2)+6)) + list(range(15*(k+3), 15*(k+3)+6))
    rows = [r for r in rows if r < n]
    Lref = np.array(L[np.ix_(rows, range(15*k, 15*k+15))], dtype=np.float64)
    Pk = P[k][:len(rows)].copy()
    for c in range(15):
        Pk[c, c] = 1.0 / Pk[c, c]
    Pk[:15] = np.tril(Pk[:15])
    err = np.abs(Pk - Lref) / (np.abs(Lref).max())
    worst.append(err.max())
    if k<3: print('kf', k, 'panel max rel err', err.max(), 'at', np.unravel_index(err.argmax(), err.shape), ' y err', np.abs(P[k][42] - y[15*k:15*k+15]).max() / np.abs(y).max())
print('panel err per kf', np.array(worst))
# emulate the back substitution from the GPU panels
d_gpu = eng.read_delta(0, 0, N)
delta = np.zeros((N + 4, 15))
for k in range(N - 1, -1, -1):
    Pk = P[k]
    s = Pk[42].copy()
    for pp in range(15, 42):
        dd = 1 if pp < 30 else (2 if pp < 36 else 3)
        a = pp - 15 if pp < 30 else (pp - 30 if pp < 36 else pp - 36)
        s -= Pk[pp] * delta[k + dd, a]
    x = np.zeros(15)
    for c in range(14, -1, -1):
        x[c] = s[c] * Pk[c, c]
        s[:c] -= Pk[c, :c] * x[c]
    delta[k] = x
rc, do = oracle.band_solve(H, g, 1e-5)
print('emulated vs oracle', np.abs(delta[:N] - do).max() / np.abs(do).max())
print('gpu vs emulated   ', np.abs(delta[:N] - d_gpu).max() / np.abs(do).max())
e = np.abs(delta[:N] - d_gpu) / np.abs(do).max()
print('per kf', e.max(axis=1))
print('per dof', e.max(axis=0))
