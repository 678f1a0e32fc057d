"""numpy prototype of the partitioned (separator) solve of the block-banded system -- pins the algebra
of K4p before the HIP kernels.  Profile: keyframe k couples to k+1 (15x15), k+2/k+3 (pose 6x6)."""
import numpy as np
rng = np.random.default_rng(0)
n, P = 50, 3
N = 15 * n
# SPD matrix with the profile: sum of factor-like J^T J
H = np.zeros((N, N)); g = rng.normal(size=N)
for k in range(n):
    J = rng.normal(size=(15, 15)); H[15*k:15*k+15, 15*k:15*k+15] += J.T @ J + 1e-3*np.eye(15)
for k in range(1, n):
    J = rng.normal(size=(15, 30)); idx = np.r_[15*(k-1):15*k, 15*k:15*k+15]
    H[np.ix_(idx, idx)] += J.T @ J
for k in range(2, n):
    for d in (1, 2, 3):
        if k - d < 0 or rng.random() < 0.3: continue
        J = rng.normal(size=(6, 12)); idx = np.r_[15*(k-d):15*(k-d)+6, 15*k:15*k+6]
        H[np.ix_(idx, idx)] += J.T @ J
lam = 1e-3
A = H + lam*np.eye(N)
ref = np.linalg.solve(A, -g)

# chunk geometry: interiors multiple of 4 except the last
def geometry(n, P):
    L = ((n - 3*(P-1)) // P) & ~3
    ch = []; i0 = 0
    for c in range(P):
        i1 = i0 + L if c < P-1 else n
        ch.append((i0, i1)); i0 = i1 + 3
    return ch
ch = geometry(n, P)
print(ch)
dof = lambda a, b: np.r_[15*a:15*b]
delta = np.zeros(N)
D = []; C = []; rhs = []
Ls = []; Vs = []; ys = []
for c, (i0, i1) in enumerate(ch):
    I = dof(i0, i1)
    Aii = A[np.ix_(I, I)]; Lc = np.linalg.cholesky(Aii); Ls.append(Lc)
    y = np.linalg.solve(Lc, -g[I]); ys.append(y)
    if c < P-1:
        Sr = dof(i1, i1+3); F = A[np.ix_(I, Sr)]           # coupling interior -> right separator
        X = np.linalg.solve(Lc, F)
        R = A[np.ix_(Sr, Sr)] - X.T @ X                    # forward remainder (own H_SS + right-side Schur)
        r = -g[Sr] - X.T @ y
        D.append(R); rhs.append(r)
    if c > 0:
        Sl = dof(i0-3, i0); E = A[np.ix_(I, Sl)]
        V = np.linalg.solve(Lc, E); Vs.append(V)           # spike
        D[c-1] -= V.T @ V; rhs[c-1] -= V.T @ y
        if c < P-1:
            C.append(-(X.T @ V))                           # cross: rows right separator, cols left separator
    else:
        Vs.append(None)
# reduced block-tridiagonal system
m = P-1
Rm = np.zeros((45*m, 45*m)); rr = np.concatenate(rhs)
for s in range(m):
    Rm[45*s:45*s+45, 45*s:45*s+45] = D[s]
for s in range(m-1):
    Rm[45*(s+1):45*(s+2), 45*s:45*s+45] = C[s]; Rm[45*s:45*s+45, 45*(s+1):45*(s+2)] = C[s].T
ds = np.linalg.solve(Rm, rr)
for s in range(m):
    i1 = ch[s][1]; delta[dof(i1, i1+3)] = ds[45*s:45*s+45]
for c, (i0, i1) in enumerate(ch):
    I = dof(i0, i1); y = ys[c].copy()
    if c > 0: y -= Vs[c] @ delta[dof(i0-3, i0)]            # y' = y - V delta_left
    if c < P-1:
        F = A[np.ix_(I, dof(i1, i1+3))]; X = np.linalg.solve(Ls[c], F)
        y -= X @ delta[dof(i1, i1+3)]                       # the ordinary band back-substitution start
    delta[I] = np.linalg.solve(Ls[c].T, y)
print("max err", np.abs(delta - ref).max(), np.abs(ref).max())
