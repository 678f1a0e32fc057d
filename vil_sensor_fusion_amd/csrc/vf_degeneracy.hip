// vf_degeneracy.hip -- K6: batched degeneracy metrics on 6x6 information / covariance matrices
// and their 3x3 translation / rotation blocks, one lane per message, everything in registers.
//
// Replaces the per-message numpy/LAPACK calls of the reference's metric library
// (vil_fusion/python/degeneracy_detection_functions.py:38-251) as driven by
// apply_degen_function (vil_fusion/python/make_prettier_graphs.py:547-576) and by the online
// node (vil_fusion/src/vil_fusion/degeneracy_detection.py:115-130), and the shipped float32
// D-optimality gate (gtsam_fusion/src/degerate_odometry_filter.cpp:29-47).
//
// Small-matrix kernels: cyclic Jacobi for symmetric eigenvalues, one-sided (Hestenes) Jacobi for
// singular values, LU with partial pivoting for det / inverse, all with compile-time indices so
// a 6x6 lives in 36 registers.  float64 and float32 instantiations (fp32 tolerance sweep).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/vilfusion.h"

extern "C" void vf_set_last_error_(const char* msg);

namespace {

#define DI __device__ __forceinline__

template <typename T> struct Lim;
template <> struct Lim<double> { static constexpr double eps = 2.220446049250313e-16; static constexpr int sweeps = 12; };
template <> struct Lim<float> { static constexpr float eps = 1.1920929e-07f; static constexpr int sweeps = 8; };

template <typename T> DI T t_sqrt(T x) { return sqrt(x); }
template <typename T> DI T t_abs(T x) { return fabs(x); }

// Reciprocal and reciprocal square root for the Jacobi rotations: the hardware approximations (v_rcp / v_rsq: 2^-23
// relative in float64, 1 ulp in float32) + ONE third-order correction step in float64 (error e^3 ~ 2^-69: full double).
// No scaling for denormals / huge arguments as the library sqrt() and '/' carry (about 20 instructions each in float64):
// the arguments here are sums of squares of matrix entries, and the kernels document their range (|entry| in 1e-150 ... 1e150).
DI double rsq_full(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}
DI float rsq_full(float x) { return __builtin_amdgcn_rsqf(x); }
DI double rcp_full(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
DI float rcp_full(float x) { return __builtin_amdgcn_rcpf(x); }

// Jacobi rotation that annihilates beta in [[a, beta], [beta, b]], alpha = (b - a) / 2:
//   t = tan(phi) = sgn(alpha) beta / (|alpha| + sqrt(alpha^2 + beta^2)),  c = 1 / sqrt(1 + t^2),  s = t c
// (the textbook theta = alpha / beta form costs two divisions and two square roots; this one a reciprocal square root, a
// reciprocal and a second reciprocal square root).  beta = 0 gives t = 0, c = 1, s = 0 without a branch.
template <typename T>
DI void jacobi_cs(T alpha, T beta, T& t, T& c, T& s) {
    const T h2 = fma(alpha, alpha, beta * beta);
    const T h = h2 * rsq_full(h2 > T(0) ? h2 : T(1));
    const T den = copysign(t_abs(alpha) + h, alpha);
    t = beta != T(0) ? beta * rcp_full(den) : T(0);
    c = rsq_full(fma(t, t, T(1)));
    s = t * c;
}

// symmetric eigenvalues, cyclic Jacobi on the UPPER TRIANGLE (a[p * N + q], p <= q; the strict lower triangle is neither
// read nor written).  A sweep ends the iteration for the whole wave once every lane's off-diagonal mass has dropped to
// rounding level, off^2 <= eps^2 * sum diag^2 (eigenvalue error ~ off^2 / gap: far below eps for any spectrum); a 6x6
// takes 5-7 sweeps, the cap Lim<T>::sweeps is a safety net.  `active`: lanes without a matrix do not hold the wave back.
template <typename T, int N>
DI void jacobi_eig(T (&a)[N * N], T (&ev)[N], bool active) {
#pragma unroll 1
    for (int sweep = 0; sweep < Lim<T>::sweeps; sweep++) {
        T off2 = T(0), d2 = T(0);
#pragma unroll
        for (int p = 0; p < N; p++) {
            d2 = fma(a[p * N + p], a[p * N + p], d2);
#pragma unroll
            for (int q = p + 1; q < N; q++) off2 = fma(a[p * N + q], a[p * N + q], off2);
        }
        const bool done = !active || !(off2 > Lim<T>::eps * Lim<T>::eps * d2);
        if (__all(done)) break;
#pragma unroll
        for (int p = 0; p < N - 1; p++)
#pragma unroll
            for (int q = p + 1; q < N; q++) {
                T t, c, s;
                const T apq = a[p * N + q];
                jacobi_cs<T>(T(0.5) * (a[q * N + q] - a[p * N + p]), apq, t, c, s);
                a[p * N + p] = fma(-t, apq, a[p * N + p]);
                a[q * N + q] = fma(t, apq, a[q * N + q]);
                a[p * N + q] = T(0);
#pragma unroll
                for (int k = 0; k < N; k++) {
                    if (k == p || k == q) continue;
                    // entry (k, p) and (k, q) of the symmetric matrix, wherever the upper triangle keeps them
                    T& xp = k < p ? a[k * N + p] : a[p * N + k];
                    T& xq = k < q ? a[k * N + q] : a[q * N + k];
                    const T vp = xp, vq = xq;
                    xp = fma(c, vp, -(s * vq));
                    xq = fma(s, vp, c * vq);
                }
            }
    }
#pragma unroll
    for (int i = 0; i < N; i++) ev[i] = a[i * N + i];
}

// Symmetric eigenvalues by Householder tridiagonalisation + implicit QL with Wilkinson shifts and deflation, all in registers
// (compile-time indices; what varies from lane to lane -- where the unreduced block ends, whether a lane has converged -- is
// carried by selects, never by an index).  The metrics need the two ends of the spectrum only, and a full cyclic Jacobi
// (5-7 sweeps of 15 rotations, every rotation touching two rows and two columns: 4 900 vector instructions per wave for a 6 x 6)
// pays for eigenvectors nobody asked for; here the N - 2 reflections cost ~250 instructions and every QL iteration is N - 1 - l
// rotations of a tridiagonal (three numbers each): ~1 800 in all.  Accuracy is that of LAPACK's tridiagonal QL -- absolute,
// eps * |A| -- which is what the reference's own numpy.linalg.eigvals / cond deliver (Jacobi's relative accuracy on graded
// matrices is not needed: tests/test_gpu_degeneracy.py holds e_opt to 1e-9 * max|A|).  The upper triangle of `a` is read;
// a is destroyed.
template <typename T, int N>
DI void sym_eig(T (&a)[N * N], T (&d)[N], bool active) {
    T e[N];          // e[k] couples k and k + 1; e[N - 1] = 0 is the sentinel of the QL loop
#pragma unroll
    for (int k = 0; k < N; k++) e[k] = T(0);
    // ---- Householder: rows i = N-1 .. 2 of the lower triangle (held as a[c * N + r], c <= r: the upper triangle of the array)
#define SA(r, c) a[((r) < (c) ? (r) : (c)) * N + ((r) < (c) ? (c) : (r))]
#pragma unroll
    for (int i = N - 1; i >= 2; i--) {
        T tail = T(0);      // what the reflection has to annihilate: row i, columns 0 .. i-2
#pragma unroll
        for (int k = 0; k < i - 1; k++) tail = fma(SA(i, k), SA(i, k), tail);
        const T xl = SA(i, i - 1);
        const T sigma = fma(xl, xl, tail);
        const bool skip = !(tail > T(0));                 // already tridiagonal here (or a lane without a matrix)
        const T nrm = sigma * rsq_full(skip ? T(1) : sigma);
        const T alpha = -copysign(nrm, xl);
        const T h = fma(-xl, alpha, sigma);                // = v.v / 2 with v = x - alpha e
        const T inv_h = skip ? T(0) : rcp_full(h);
        T v[N], pv[N];
#pragma unroll
        for (int k = 0; k < i; k++) v[k] = skip ? T(0) : (k == i - 1 ? xl - alpha : SA(i, k));
        T f = T(0);
#pragma unroll
        for (int j = 0; j < i; j++) {
            T g = T(0);
#pragma unroll
            for (int k = 0; k < i; k++) g = fma(SA(j, k), v[k], g);
            pv[j] = g * inv_h;
            f = fma(pv[j], v[j], f);
        }
        const T hh = T(0.5) * f * inv_h;
#pragma unroll
        for (int j = 0; j < i; j++) pv[j] = fma(-hh, v[j], pv[j]);
#pragma unroll
        for (int j = 0; j < i; j++)
#pragma unroll
            for (int k = 0; k <= j; k++) SA(j, k) = SA(j, k) - fma(v[j], pv[k], pv[j] * v[k]);
        e[i - 1] = skip ? xl : alpha;
    }
    if (N >= 2) e[0] = SA(1, 0);
#pragma unroll
    for (int k = 0; k < N; k++) d[k] = SA(k, k);
#undef SA
    // ---- implicit QL (the classical tqli recurrence), eigenvalue by eigenvalue
#pragma unroll
    for (int l = 0; l < N - 1; l++) {
#pragma unroll 1
        for (int iter = 0; iter < 40; iter++) {
            int m = N - 1;                                  // the smallest m >= l whose coupling to m + 1 is negligible
#pragma unroll
            for (int k = N - 2; k >= l; k--) {
                const T dd = t_abs(d[k]) + t_abs(d[k + 1]);
                m = (t_abs(e[k]) <= Lim<T>::eps * dd) ? k : m;
            }
            const bool work = active && m != l;
            if (!__any(work)) break;
            const T el = work ? e[l] : T(1);
            T g = (d[l + 1] - d[l]) * rcp_full(el + el);
            T r2 = fma(g, g, T(1));
            T r = r2 * rsq_full(r2);
            T dm = d[N - 1];
#pragma unroll
            for (int k = N - 2; k > l; k--) dm = (m == k) ? d[k] : dm;
            g = dm - d[l] + el * rcp_full(g + copysign(r, g));
            T sn = T(1), cs = T(1), pp = T(0);
            bool brk = false;
            // (the lanes that take no part in a rotation are masked off by the branch, not by selects on every result: a
            // rotation is ~26 vector instructions this way, ~40 with selects)
#pragma unroll
            for (int i = N - 2; i >= l; i--) {
                if (work && i < m && !brk) {
                    const T f = sn * e[i], b = cs * e[i];
                    const T rr = fma(f, f, g * g);
                    if (!(rr > T(0))) {              // (underflow: the block has split here)
                        e[i + 1] = T(0);
                        d[i + 1] -= pp;
                        brk = true;
                    } else {
                        const T inv = rsq_full(rr);
                        e[i + 1] = rr * inv;
                        sn = f * inv;
                        cs = g * inv;
                        const T gg = d[i + 1] - pp;
                        const T rot = fma(d[i] - gg, sn, T(2) * cs * b);
                        pp = sn * rot;
                        d[i + 1] = gg + pp;
                        g = fma(cs, rot, -b);
                    }
                }
            }
            if (work && !brk) { d[l] -= pp; e[l] = g; }
#pragma unroll
            for (int k = l; k < N; k++) e[k] = (work && m == k) ? T(0) : e[k];
        }
    }
}

// singular values of a general matrix, one-sided (Hestenes) Jacobi on the columns; a sweep in which no lane of the wave
// rotated ends the iteration
template <typename T, int N>
DI void jacobi_svd(T (&a)[N * N], T (&sv)[N], bool active) {
#pragma unroll 1
    for (int sweep = 0; sweep < Lim<T>::sweeps; sweep++) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < N - 1; p++)
#pragma unroll
            for (int q = p + 1; q < N; q++) {
                T al = 0, be = 0, ga = 0;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    al = fma(a[k * N + p], a[k * N + p], al);
                    be = fma(a[k * N + q], a[k * N + q], be);
                    ga = fma(a[k * N + p], a[k * N + q], ga);
                }
                constexpr T tol = Lim<T>::eps * T(0.01);
                const bool need = ga * ga > tol * tol * al * be;
                rotated |= need;
                T t, c, s;
                jacobi_cs<T>(T(0.5) * (be - al), need ? ga : T(0), t, c, s);
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const T akp = a[k * N + p], akq = a[k * N + q];
                    a[k * N + p] = fma(c, akp, -(s * akq));
                    a[k * N + q] = fma(s, akp, c * akq);
                }
            }
        if (__all(!active || !rotated)) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        T s = 0;
#pragma unroll
        for (int k = 0; k < N; k++) s = fma(a[k * N + i], a[k * N + i], s);
        sv[i] = t_sqrt(s);
    }
}

// LU with partial pivoting (row swaps by predicated moves): log|det| as numpy.linalg.slogdet()[1]
template <typename T, int N>
DI T lu_logabsdet(const T (&m)[N * N]) {
    T a[N * N];
#pragma unroll
    for (int i = 0; i < N * N; i++) a[i] = m[i];
    T logabs = T(0);
#pragma unroll
    for (int c = 0; c < N; c++) {
        int piv = c;
        T best = t_abs(a[c * N + c]);
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const T v = t_abs(a[r * N + c]);
            if (v > best) { best = v; piv = r; }
        }
#pragma unroll
        for (int r = c + 1; r < N; r++)
            if (piv == r) {
#pragma unroll
                for (int k = 0; k < N; k++) { const T t = a[c * N + k]; a[c * N + k] = a[r * N + k]; a[r * N + k] = t; }
            }
        const T d = a[c * N + c];
        logabs += log(t_abs(d));
        const T inv = T(1) / d;
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const T f = a[r * N + c] * inv;
#pragma unroll
            for (int k = c + 1; k < N; k++) a[r * N + k] = fma(-f, a[c * N + k], a[r * N + k]);
        }
    }
    return logabs;
}

// plain determinant with the swap parity applied
template <typename T, int N>
DI T lu_det(const T (&m)[N * N]) {
    T a[N * N];
#pragma unroll
    for (int i = 0; i < N * N; i++) a[i] = m[i];
    T det = T(1);
#pragma unroll
    for (int c = 0; c < N; c++) {
        int piv = c;
        T best = t_abs(a[c * N + c]);
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const T v = t_abs(a[r * N + c]);
            if (v > best) { best = v; piv = r; }
        }
#pragma unroll
        for (int r = c + 1; r < N; r++)
            if (piv == r) {
#pragma unroll
                for (int k = 0; k < N; k++) { const T t = a[c * N + k]; a[c * N + k] = a[r * N + k]; a[r * N + k] = t; }
                det = -det;
            }
        const T d = a[c * N + c];
        det *= d;
        const T inv = T(1) / d;
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const T f = a[r * N + c] * inv;
#pragma unroll
            for (int k = c + 1; k < N; k++) a[r * N + k] = fma(-f, a[c * N + k], a[r * N + k]);
        }
    }
    return det;
}

// inverse by Gauss-Jordan with partial pivoting
template <typename T, int N>
DI void gj_inverse(const T (&m)[N * N], T (&inv)[N * N]) {
    T a[N * N];
#pragma unroll
    for (int i = 0; i < N * N; i++) { a[i] = m[i]; inv[i] = T(0); }
#pragma unroll
    for (int i = 0; i < N; i++) inv[i * N + i] = T(1);
#pragma unroll
    for (int c = 0; c < N; c++) {
        int piv = c;
        T best = t_abs(a[c * N + c]);
#pragma unroll
        for (int r = c + 1; r < N; r++) {
            const T v = t_abs(a[r * N + c]);
            if (v > best) { best = v; piv = r; }
        }
#pragma unroll
        for (int r = c + 1; r < N; r++)
            if (piv == r) {
#pragma unroll
                for (int k = 0; k < N; k++) {
                    T t = a[c * N + k]; a[c * N + k] = a[r * N + k]; a[r * N + k] = t;
                    t = inv[c * N + k]; inv[c * N + k] = inv[r * N + k]; inv[r * N + k] = t;
                }
            }
        const T ip = T(1) / a[c * N + c];
#pragma unroll
        for (int k = 0; k < N; k++) { a[c * N + k] *= ip; inv[c * N + k] *= ip; }
#pragma unroll
        for (int r = 0; r < N; r++)
            if (r != c) {
                const T f = a[r * N + c];
#pragma unroll
                for (int k = 0; k < N; k++) {
                    a[r * N + k] = fma(-f, a[c * N + k], a[r * N + k]);
                    inv[r * N + k] = fma(-f, inv[c * N + k], inv[r * N + k]);
                }
            }
    }
}

template <typename T, int N>
DI void matmul(const T (&a)[N * N], const T (&b)[N * N], T (&c)[N * N]) {
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) {
            T s = 0;
#pragma unroll
            for (int k = 0; k < N; k++) s = fma(a[i * N + k], b[k * N + j], s);
            c[i * N + j] = s;
        }
}

// eigenvalues of now * prev^-1 for SPD prev: similar to L^-1 now L^-T with prev = L L^T
template <typename T, int N>
DI void ratio_eig(const T (&now)[N * N], const T (&prev)[N * N], T (&ev)[N], bool& ok, bool active) {
    T L[N * N];
#pragma unroll
    for (int i = 0; i < N * N; i++) L[i] = T(0);
    ok = true;
#pragma unroll
    for (int j = 0; j < N; j++) {
        T s = prev[j * N + j];
#pragma unroll
        for (int k = 0; k < j; k++) s = fma(-L[j * N + k], L[j * N + k], s);
        if (!(s > T(0))) { ok = false; s = T(1); }
        const T ljj = t_sqrt(s);
        L[j * N + j] = ljj;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            T v = T(0.5) * (prev[i * N + j] + prev[j * N + i]);
#pragma unroll
            for (int k = 0; k < j; k++) v = fma(-L[i * N + k], L[j * N + k], v);
            L[i * N + j] = v / ljj;
        }
    }
    // Y = L^-1 now (forward substitution on each column), S = Y L^-T (same on rows)
    T Y[N * N], S[N * N];
#pragma unroll
    for (int c = 0; c < N; c++)
#pragma unroll
        for (int r = 0; r < N; r++) {
            T v = T(0.5) * (now[r * N + c] + now[c * N + r]);
#pragma unroll
            for (int k = 0; k < r; k++) v = fma(-L[r * N + k], Y[k * N + c], v);
            Y[r * N + c] = v / L[r * N + r];
        }
#pragma unroll
    for (int r = 0; r < N; r++)
#pragma unroll
        for (int c = 0; c < N; c++) {
            T v = Y[r * N + c];
#pragma unroll
            for (int k = 0; k < c; k++) v = fma(-L[c * N + k], S[r * N + k], v);
            S[r * N + c] = v / L[c * N + c];
        }
#pragma unroll
    for (int r = 0; r < N; r++)
#pragma unroll
        for (int c = r + 1; c < N; c++) { const T m = T(0.5) * (S[r * N + c] + S[c * N + r]); S[r * N + c] = m; S[c * N + r] = m; }
    sym_eig<T, N>(S, ev, active);
}

enum Metric { D_OPT, D_OPT_RATIO, A_OPT, A_OPT_RATIO, E_OPT, E_OPT_RATIO, MAX_EIGEN, MAX_EIGEN_RATIO, JENSEN_BREGMAN,
              CORR_DIST, KULLBACK_LEIBLER, NORM_FRO, NORM_FRO_RATIO, NORM_NUC, NORM_NUC_RATIO, NORM_1, NORM_1_RATIO,
              NORM_2, NORM_2_RATIO, COND_NUMBER, DIFF_ENTROPY, N_METRICS,
              SPECTRUM = N_METRICS };     // (not a metric of the reference: e_opt, max_eigen and condition_number of one eigen-solve, vf_degeneracy_spectrum_batch)

template <typename T, int N>
DI T norm1(const T (&a)[N * N]) {
    T best = 0;
#pragma unroll
    for (int c = 0; c < N; c++) {
        T s = 0;
#pragma unroll
        for (int r = 0; r < N; r++) s += t_abs(a[r * N + c]);
        best = s > best ? s : best;
    }
    return best;
}
template <typename T, int N>
DI T normfro(const T (&a)[N * N]) {
    T s = 0;
#pragma unroll
    for (int i = 0; i < N * N; i++) s = fma(a[i], a[i], s);
    return t_sqrt(s);
}

// Which metrics read the previous message's matrix (the ratio / divergence family, make_prettier_graphs.py:565-574)
__host__ __device__ constexpr bool metric_needs_prev(int m) {
    return m == D_OPT_RATIO || m == A_OPT_RATIO || m == E_OPT_RATIO || m == MAX_EIGEN_RATIO || m == JENSEN_BREGMAN || m == CORR_DIST ||
           m == KULLBACK_LEIBLER || m == NORM_FRO_RATIO || m == NORM_NUC_RATIO || m == NORM_1_RATIO || m == NORM_2_RATIO;
}
constexpr int MSTRIDE = 37;   // LDS stride of one 6x6 (36 + 1 pad: 64 lanes reading entry e of their own matrix hit 32 / 64 distinct banks)

// mats: (T,6,6) row-major, pose (T,6) or null; off = 0 (all / trans) or 3 (rot); out[0] = 0.
// One wave per 64 consecutive messages.  The wave's matrices (and the one in front of them, for the metrics that compare
// with the previous message) are ONE contiguous stretch of HBM: it is copied to LDS with 16-byte loads, fully
// coalesced, and every lane then picks its own matrix out of LDS (lane = message reads of 288-byte records straight
// from HBM touch 64 cache lines per load instruction).  METRIC is a template parameter: a metric's kernel contains
// only its own arithmetic and loads the previous matrix only if it uses it.
template <typename T, int N, int METRIC>
__global__ void __launch_bounds__(64) k_degeneracy(const T* __restrict__ mats, const T* __restrict__ pose, int count, int off,
                                                   T* __restrict__ out, T* __restrict__ out2 = nullptr, T* __restrict__ out3 = nullptr) {
    // staged in two halves of 32 messages (lanes 0-31 pick theirs up after the first, 32-63 after the second): 9.8 KB of LDS
    // per wave instead of 19.2, i.e. four waves per SIMD instead of two for the Jacobi kernels
    __shared__ T lds[33 * MSTRIDE];
    const int lane = threadIdx.x, base = blockIdx.x * 64;
    const int i = base + lane;
    constexpr bool PREV = metric_needs_prev(METRIC);
    const bool active = i > 0 && i < count;
    T now[N * N], prev[N * N];
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int hb = base + 32 * half;                                  // first message of this half
        if (half) __syncthreads();
        if (hb < count) {
            const int first = (PREV && hb > 0) ? hb - 1 : hb;            // first matrix staged, local index first - hb + 1
            const int last = hb + 32 < count ? hb + 32 : count;
            constexpr int V = 16 / (int)sizeof(T);                        // elements per 16-byte load (36 is a multiple of both)
            const int nvec = (last - first) * 36 / V;
            using vec_t = T __attribute__((ext_vector_type(V)));
            const vec_t* src = reinterpret_cast<const vec_t*>(mats + (size_t)first * 36);
            for (int v = lane; v < nvec; v += 64) {
                const vec_t x = __builtin_nontemporal_load(src + v);
                const int e = v * V, m = e / 36, r = e - m * 36;
                T* dst = lds + (m + first - hb + 1) * MSTRIDE + r;
#pragma unroll
                for (int k = 0; k < V; k++) dst[k] = x[k];
            }
        }
        __syncthreads();
        if ((lane >> 5) == half) {
            const int l = lane & 31;
            const T* mn = lds + (l + 1) * MSTRIDE + off * 7;
            const T* mp = lds + l * MSTRIDE + off * 7;
#pragma unroll
            for (int r = 0; r < N; r++)
#pragma unroll
                for (int c = 0; c < N; c++) {
                    now[r * N + c] = active ? mn[r * 6 + c] : T(r == c);
                    prev[r * N + c] = (PREV && active) ? mp[r * 6 + c] : T(r == c);
                }
        }
    }
    const T nan = T(NAN);
    T y = nan;
    constexpr bool is_ratio = METRIC == D_OPT_RATIO || METRIC == A_OPT_RATIO || METRIC == NORM_FRO_RATIO ||
                              METRIC == NORM_NUC_RATIO || METRIC == NORM_1_RATIO || METRIC == NORM_2_RATIO;
    T ratio[N * N];
    if (is_ratio) {
        T pinv[N * N];
        gj_inverse<T, N>(prev, pinv);
        matmul<T, N>(now, pinv, ratio);
    }
    switch (METRIC) {
        case D_OPT: case D_OPT_RATIO: {
            const T la = (METRIC == D_OPT) ? lu_logabsdet<T, N>(now) : lu_logabsdet<T, N>(ratio);
            y = exp(la / T(N));
        } break;
        case A_OPT: case A_OPT_RATIO: {
            T s = 0;
#pragma unroll
            for (int k = 0; k < N; k++) s += (METRIC == A_OPT) ? now[k * N + k] : ratio[k * N + k];
            y = s;
        } break;
        case SPECTRUM: {
            // the three metrics that read the ends of the spectrum, from ONE eigen-solve of the symmetric part (e_opt and max_eigen
            // are defined on it; condition_number too whenever the matrix is symmetric to rounding, see COND_NUMBER below --
            // other matrices get NaN there and the caller asks for condition_number by itself)
            T s[N * N], ev[N], asym = T(0), big = T(0);
#pragma unroll
            for (int r = 0; r < N; r++)
#pragma unroll
                for (int c = 0; c < N; c++) {
                    s[r * N + c] = T(0.5) * (now[r * N + c] + now[c * N + r]);
                    big = t_abs(now[r * N + c]) > big ? t_abs(now[r * N + c]) : big;
                    if (c > r) { const T dsy = t_abs(now[r * N + c] - now[c * N + r]); asym = dsy > asym ? dsy : asym; }
                }
            sym_eig<T, N>(s, ev, active);
            T lo = ev[0], hi = ev[0], alo = t_abs(ev[0]), ahi = t_abs(ev[0]);
#pragma unroll
            for (int k = 1; k < N; k++) {
                lo = ev[k] < lo ? ev[k] : lo; hi = ev[k] > hi ? ev[k] : hi;
                const T x = t_abs(ev[k]); alo = x < alo ? x : alo; ahi = x > ahi ? x : ahi;
            }
            y = lo;
            if (i == 0) { out2[0] = T(0); out3[0] = T(0); }
            else if (i < count) { out2[i] = hi; out3[i] = asym <= T(8) * Lim<T>::eps * big ? -(ahi / alo) : nan; }
        } break;
        case E_OPT: case MAX_EIGEN: {
            T s[N * N], ev[N];
#pragma unroll
            for (int r = 0; r < N; r++)
#pragma unroll
                for (int c = 0; c < N; c++) s[r * N + c] = T(0.5) * (now[r * N + c] + now[c * N + r]);
            sym_eig<T, N>(s, ev, active);
            T lo = ev[0], hi = ev[0];
#pragma unroll
            for (int k = 1; k < N; k++) { lo = ev[k] < lo ? ev[k] : lo; hi = ev[k] > hi ? ev[k] : hi; }
            y = METRIC == E_OPT ? lo : hi;
        } break;
        case E_OPT_RATIO: case MAX_EIGEN_RATIO: {
            T ev[N];
            bool ok;
            ratio_eig<T, N>(now, prev, ev, ok, active);
            T lo = ev[0], hi = ev[0];
#pragma unroll
            for (int k = 1; k < N; k++) { lo = ev[k] < lo ? ev[k] : lo; hi = ev[k] > hi ? ev[k] : hi; }
            y = ok ? (METRIC == E_OPT_RATIO ? lo : hi) : nan;
        } break;
        case JENSEN_BREGMAN: {
            T avg[N * N], prod[N * N];
#pragma unroll
            for (int k = 0; k < N * N; k++) avg[k] = (now[k] + prev[k]) / T(2);
            matmul<T, N>(now, prev, prod);
            y = lu_logabsdet<T, N>(avg) - T(0.5) * lu_det<T, N>(prod);
        } break;
        case CORR_DIST: {   // elementwise-product quirk: only the diagonal of the "correlation" survives
            T tr = 0, fx = 0, fy = 0;
            bool ok = true;
#pragma unroll
            for (int k = 0; k < N; k++) {
                const T dn = t_sqrt(now[k * N + k]), dp = t_sqrt(prev[k * N + k]);
                if (!(dn > T(0)) || !(dp > T(0))) ok = false;
                const T cn = (T(1) / dn) * now[k * N + k] * (T(1) / dn), cp = (T(1) / dp) * prev[k * N + k] * (T(1) / dp);
                tr = fma(cn, cp, tr);
                fx = fma(cn, cn, fx);
                fy = fma(cp, cp, fy);
            }
            y = ok ? T(1) - tr / (t_sqrt(fx) * t_sqrt(fy)) : nan;
        } break;
        case KULLBACK_LEIBLER: {   // E1 = prev, E2 = now
            T e2i[N * N], m[N * N];
            gj_inverse<T, N>(now, e2i);
            matmul<T, N>(e2i, prev, m);
            T a = 0;
#pragma unroll
            for (int k = 0; k < N; k++) a += m[k * N + k] - T(1);
            T du[N], b = 0;
#pragma unroll
            for (int k = 0; k < N; k++) du[k] = (pose && active) ? pose[(size_t)(i - 1) * 6 + off + k] - pose[(size_t)i * 6 + off + k] : T(0);
#pragma unroll
            for (int r = 0; r < N; r++) {
                T s = 0;
#pragma unroll
                for (int c = 0; c < N; c++) s = fma(e2i[r * N + c], du[c], s);
                b = fma(du[r], s, b);
            }
            const T c = log(t_abs(lu_det<T, N>(now)) / t_abs(lu_det<T, N>(prev)));
            y = T(0.5) * (a + b + c);
        } break;
        case NORM_FRO: y = normfro<T, N>(now); break;
        case NORM_FRO_RATIO: y = normfro<T, N>(ratio); break;
        case NORM_1: y = norm1<T, N>(now); break;
        case NORM_1_RATIO: y = norm1<T, N>(ratio); break;
        case NORM_NUC: case NORM_2: case COND_NUMBER: case NORM_NUC_RATIO: case NORM_2_RATIO: {
            T w[N * N], sv[N];
            constexpr bool r = METRIC == NORM_NUC_RATIO || METRIC == NORM_2_RATIO;
            if (METRIC == NORM_NUC || METRIC == NORM_2 || METRIC == COND_NUMBER) {
                // An information matrix is symmetric, and the singular values of a symmetric matrix are the moduli of its
                // eigenvalues: sigma_max / sigma_min, the sum and the maximum come from the eigen-solve e_opt / max_eigen use, at a
                // seventh of the cost of a Jacobi SVD.  "Symmetric" = to within rounding of the entries (an asymmetry of that
                // size moves a singular value by as much: what the SVD's own rounding does); a wave with a lane that holds
                // anything else takes the general path below.
                T asym = T(0), big = T(0);
#pragma unroll
                for (int rr = 0; rr < N; rr++)
#pragma unroll
                    for (int cc = 0; cc < N; cc++) {
                        big = t_abs(now[rr * N + cc]) > big ? t_abs(now[rr * N + cc]) : big;
                        if (cc > rr) { const T dsy = t_abs(now[rr * N + cc] - now[cc * N + rr]); asym = dsy > asym ? dsy : asym; }
                    }
                if (__all(!active || asym <= T(8) * Lim<T>::eps * big)) {
                    T sy[N * N], ev[N];
#pragma unroll
                    for (int rr = 0; rr < N; rr++)
#pragma unroll
                        for (int cc = 0; cc < N; cc++) sy[rr * N + cc] = T(0.5) * (now[rr * N + cc] + now[cc * N + rr]);
                    sym_eig<T, N>(sy, ev, active);
                    T lo = t_abs(ev[0]), hi = t_abs(ev[0]), sum = T(0);
#pragma unroll
                    for (int k = 0; k < N; k++) { const T x = t_abs(ev[k]); lo = x < lo ? x : lo; hi = x > hi ? x : hi; sum += x; }
                    y = METRIC == NORM_NUC ? sum : (METRIC == COND_NUMBER ? -(hi / lo) : hi);
                    break;
                }
            }
#pragma unroll
            for (int k = 0; k < N * N; k++) w[k] = r ? ratio[k] : now[k];
            jacobi_svd<T, N>(w, sv, active);
            T lo = sv[0], hi = sv[0], sum = 0;
#pragma unroll
            for (int k = 0; k < N; k++) { lo = sv[k] < lo ? sv[k] : lo; hi = sv[k] > hi ? sv[k] : hi; sum += sv[k]; }
            y = (METRIC == NORM_NUC || METRIC == NORM_NUC_RATIO) ? sum : ((METRIC == COND_NUMBER) ? -(hi / lo) : hi);
        } break;
        case DIFF_ENTROPY: {
            const T x = pow(T(2.0 * 3.141592653589793 * 2.718281828459045), T(N));
            const T d = x * lu_det<T, N>(now);
            y = d > T(0) ? T(0.5) * log(d) : nan;
        } break;
        default: break;
    }
    if (i == 0) out[0] = T(0);            // make_prettier_graphs.py:562-563
    else if (i < count) out[i] = y;
}

template <typename T, int N>
void launch_degeneracy(int metric, dim3 grid, const T* m, const T* p, int count, int off, T* o) {
#define VF_K6_CASE(M) case M: hipLaunchKernelGGL((k_degeneracy<T, N, M>), grid, dim3(64), 0, 0, m, p, count, off, o); break;
    switch (metric) {
        VF_K6_CASE(0) VF_K6_CASE(1) VF_K6_CASE(2) VF_K6_CASE(3) VF_K6_CASE(4) VF_K6_CASE(5) VF_K6_CASE(6) VF_K6_CASE(7)
        VF_K6_CASE(8) VF_K6_CASE(9) VF_K6_CASE(10) VF_K6_CASE(11) VF_K6_CASE(12) VF_K6_CASE(13) VF_K6_CASE(14) VF_K6_CASE(15)
        VF_K6_CASE(16) VF_K6_CASE(17) VF_K6_CASE(18) VF_K6_CASE(19) VF_K6_CASE(20)
        default: break;
    }
#undef VF_K6_CASE
}
static_assert(N_METRICS == 21, "launch_degeneracy lists every metric");

template <typename T>
int run_spectrum(const void* mats, int count, int subset, void* o_min, void* o_max, void* o_cond, int reps, float* kernel_ms);

// degerate_odometry_filter.cpp:29-47 (float32): hessian (row-major floats) copied into a
// column-major Eigen matrix, rotation = block(3,3), translation = block(0,0), log(det)
__global__ void k_dopt_filter(const float* __restrict__ h, int count, float rot_thr, float trans_thr,
                              float* __restrict__ rot, float* __restrict__ trans, unsigned char* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float* m = h + (size_t)i * 36;
    auto det3 = [&](int o) {   // Eigen(r,c) = m[c*6 + r]
        auto E = [&](int r, int c) { return m[(o + c) * 6 + o + r]; };
        return E(0, 0) * (E(1, 1) * E(2, 2) - E(1, 2) * E(2, 1)) - E(0, 1) * (E(1, 0) * E(2, 2) - E(1, 2) * E(2, 0)) +
               E(0, 2) * (E(1, 0) * E(2, 1) - E(1, 1) * E(2, 0));
    };
    const float r = logf(det3(3)), t = logf(det3(0));
    rot[i] = r;
    trans[i] = t;
    keep[i] = !(r < rot_thr || t < trans_thr);
}

int derr(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    vf_set_last_error_(buf);
    return code;
}
#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) return derr(VF_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

template <typename T>
int run_batch(const void* mats, const void* pose, int count, int subset, int metric, void* out, int reps, float* kernel_ms) {
    T *d_m = nullptr, *d_p = nullptr, *d_o = nullptr;
    const size_t mb = (size_t)count * 36 * sizeof(T), pb = (size_t)count * 6 * sizeof(T), ob = (size_t)count * sizeof(T);
    HIPCHK(hipMalloc((void**)&d_m, mb));
    HIPCHK(hipMalloc((void**)&d_o, ob));
    HIPCHK(hipMemcpy(d_m, mats, mb, hipMemcpyHostToDevice));
    if (pose) {
        HIPCHK(hipMalloc((void**)&d_p, pb));
        HIPCHK(hipMemcpy(d_p, pose, pb, hipMemcpyHostToDevice));
    }
    const int off = subset == 2 ? 3 : 0;
    const dim3 grid((count + 63) / 64);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (subset == 0) launch_degeneracy<T, 6>(metric, grid, d_m, d_p, count, off, d_o);
        else launch_degeneracy<T, 3>(metric, grid, d_m, d_p, count, off, d_o);
    };
    launch();
    HIPCHK(hipDeviceSynchronize());
    if (kernel_ms && reps > 0) {
        HIPCHK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) launch();
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        *kernel_ms = ms / reps;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, d_o, ob, hipMemcpyDeviceToHost));
    (void)hipFree(d_m); (void)hipFree(d_o);
    if (d_p) (void)hipFree(d_p);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return VF_OK;
}

template <typename T>
int run_spectrum(const void* mats, int count, int subset, void* o_min, void* o_max, void* o_cond, int reps, float* kernel_ms) {
    T *d_m = nullptr, *d_o = nullptr;
    const size_t mb = (size_t)count * 36 * sizeof(T), ob = (size_t)count * sizeof(T);
    HIPCHK(hipMalloc((void**)&d_m, mb));
    HIPCHK(hipMalloc((void**)&d_o, 3 * ob));
    HIPCHK(hipMemcpy(d_m, mats, mb, hipMemcpyHostToDevice));
    const int off = subset == 2 ? 3 : 0;
    const dim3 grid((count + 63) / 64);
    auto launch = [&]() {
        if (subset == 0) hipLaunchKernelGGL((k_degeneracy<T, 6, SPECTRUM>), grid, dim3(64), 0, 0, d_m, (const T*)nullptr, count, off, d_o, d_o + count, d_o + 2 * (size_t)count);
        else hipLaunchKernelGGL((k_degeneracy<T, 3, SPECTRUM>), grid, dim3(64), 0, 0, d_m, (const T*)nullptr, count, off, d_o, d_o + count, d_o + 2 * (size_t)count);
    };
    launch();
    HIPCHK(hipDeviceSynchronize());
    if (kernel_ms && reps > 0) {
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) launch();
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        *kernel_ms = ms / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(o_min, d_o, ob, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(o_max, d_o + count, ob, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(o_cond, d_o + 2 * (size_t)count, ob, hipMemcpyDeviceToHost));
    (void)hipFree(d_m); (void)hipFree(d_o);
    return VF_OK;
}

}  // namespace

extern "C" {

int vf_degeneracy_spectrum_batch(const void* mats, int count, int dtype, int subset, void* e_opt, void* max_eigen, void* condition_number,
                                 int reps, float* kernel_ms) {
    if (!mats || !e_opt || !max_eigen || !condition_number || count < 0) return derr(VF_ERR_INVALID, "null argument");
    if (subset < 0 || subset > 2) return derr(VF_ERR_INVALID, "subset must be 0 (all), 1 (trans) or 2 (rot)");
    if (dtype != 0 && dtype != 1) return derr(VF_ERR_INVALID, "dtype must be 0 (f64) or 1 (f32)");
    if (count == 0) return VF_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return derr(VF_ERR_NO_DEVICE, "no HIP device visible; libvilfusion has no CPU path");
    return dtype == 0 ? run_spectrum<double>(mats, count, subset, e_opt, max_eigen, condition_number, reps, kernel_ms)
                      : run_spectrum<float>(mats, count, subset, e_opt, max_eigen, condition_number, reps, kernel_ms);
}

int vf_degeneracy_batch(const void* mats, const void* pose, int count, int dtype, int subset, int metric, void* out,
                        int reps, float* kernel_ms) {
    if (!mats || !out || count < 0) return derr(VF_ERR_INVALID, "null argument");
    if (metric < 0 || metric >= N_METRICS) return derr(VF_ERR_INVALID, "unknown metric %d", metric);
    if (subset < 0 || subset > 2) return derr(VF_ERR_INVALID, "subset must be 0 (all), 1 (trans) or 2 (rot)");
    if (dtype != 0 && dtype != 1) return derr(VF_ERR_INVALID, "dtype must be 0 (f64) or 1 (f32)");
    if (metric == KULLBACK_LEIBLER && !pose) return derr(VF_ERR_INVALID, "kullback_leibler needs poses");
    if (count == 0) return VF_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return derr(VF_ERR_NO_DEVICE, "no HIP device visible; libvilfusion has no CPU path");
    return dtype == 0 ? run_batch<double>(mats, pose, count, subset, metric, out, reps, kernel_ms)
                      : run_batch<float>(mats, pose, count, subset, metric, out, reps, kernel_ms);
}

int vf_dopt_filter_f32(const float* hessians, int count, float rot_thr, float trans_thr, float* rot_dopt,
                       float* trans_dopt, unsigned char* keep) {
    if (!hessians || !rot_dopt || !trans_dopt || !keep || count < 0) return derr(VF_ERR_INVALID, "null argument");
    if (count == 0) return VF_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return derr(VF_ERR_NO_DEVICE, "no HIP device visible; libvilfusion has no CPU path");
    float *d_h = nullptr, *d_r = nullptr, *d_t = nullptr;
    unsigned char* d_k = nullptr;
    HIPCHK(hipMalloc((void**)&d_h, (size_t)count * 36 * sizeof(float)));
    HIPCHK(hipMalloc((void**)&d_r, count * sizeof(float)));
    HIPCHK(hipMalloc((void**)&d_t, count * sizeof(float)));
    HIPCHK(hipMalloc((void**)&d_k, count));
    HIPCHK(hipMemcpy(d_h, hessians, (size_t)count * 36 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dopt_filter, dim3((count + 255) / 256), dim3(256), 0, 0, d_h, count, rot_thr, trans_thr, d_r, d_t, d_k);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(rot_dopt, d_r, count * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(trans_dopt, d_t, count * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(keep, d_k, count, hipMemcpyDeviceToHost));
    (void)hipFree(d_h); (void)hipFree(d_r); (void)hipFree(d_t); (void)hipFree(d_k);
    return VF_OK;
}

}  // extern "C"
