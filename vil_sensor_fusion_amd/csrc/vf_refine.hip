// vf_refine.hip -- the refined solve: conjugate gradients on the normal equations with the operator applied THROUGH J.
//
// Why.  The reference factorises by QR (GraphManager.cpp:38 `factorization = ISAM2Params::QR`); the device forms the banded
// normal equations H = J^T J and factorises them by Cholesky.  An n-keyframe chain of combined-IMU factors is a double
// integrator: its softest modes have curvature ~ stiffness / n^4, so cond(H) ~ 1e11 at n = 1 000 and beyond 1e19 at
// n = 10 000 (BASELINE configs[4]).  There the Cholesky factor of the float64 H has the soft eigenvalues wrong by orders of
// magnitude: undamped Gauss-Newton by normal equations creeps (measured on the CPU oracle: 4.4 m from the QR optimum after 8
// steps, still 0.12 m after 6 steps with an 80-bit H and factor), where QR of the whitened Jacobian (cond ~ 1e9) takes 4.
// The Jacobian itself is accurate: J v costs eps |J| |v|, and J^T (J v) inherits cond(J), not cond(J)^2.  So the Cholesky
// factor M = L L^T (whatever form the engine's K4 has: sweep, partitioned, time-sharded) is kept as a PRECONDITIONER and
// the step is the conjugate-gradient solution of (J^T J + lambda I) d = -J^T r with A p evaluated as J^T (J p) + lambda p
// (k_jv, k_jtu below): M^-1 A is the identity on all but the handful of soft modes M gets wrong, and CG removes one such
// outlier per iteration -- 8 to 10 iterations reach the accuracy of the QR step (DESIGN.md "Refined solve").  Textbook name:
// CGLS preconditioned by the Cholesky factor of the normal equations (Bjorck, Numerical Methods for Least Squares Problems, 7.4).
//
// Every vector is increment-shaped ([G][15], a window's keyframes [lo, hi)); scalars are per window; every sum is formed in one
// fixed order (64 partial sums per window, added by one lane), so a solve is bitwise reproducible and identical on every rank of a
// time-sharded window.
#include "vf_kernels.hpp"
#include "vf_jstream.hpp"

namespace vf {

#define VF_DI __device__ __forceinline__

// did the plain solve of window w fail (normal equations not positive definite)?  On a time-sharded engine the other ranks'
// flags arrive, summed, behind the increments (k_mask_delta + the all-reduce)
VF_DI bool solve_failed(const View& v, int w) {
    return v.fail[w] != 0 || (v.sh_G > 1 && v.delta[(size_t)v.G * 15 + w] > 0.0);
}
VF_DI bool refine_off(const View& v, const Refine& q, int w) {
    return q.stop[w] != 0 || (v.stop_on && v.done[w]) || v.hi[w] - v.lo[w] <= 0;
}
// GTSAM column of the 15x30 Jacobian -> tangent component of its keyframe (inverse of imu_col)
__host__ __device__ constexpr int col_comp(int col) { return col < 18 ? col % 9 : 9 + (col - 18) % 6; }

// u = J p, one lane per factor slot: the IMU factor k-1 -> k (15 rows), the between factor a -> k (6 rows) and, in the lane
// of its keyframe, the prior of GraphManager.cpp:27-35 (15 rows), all at the CURRENT linearisation (buffer sel).
__global__ void __launch_bounds__(256) k_jv(View v, Refine q, const double* __restrict__ p) {
    const long gk = (long)blockIdx.x * 256 + threadIdx.x;
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    const int lo = v.lo[w], hi = v.hi[w], b = v.sel[w];
    if (k < lo || k >= hi || refine_off(v, q, w)) return;
    double pj[15];
#pragma unroll
    for (int c = 0; c < 15; c++) pj[c] = p[(size_t)gk * 15 + c];
    if (v.prior_k[w] == k) {
        const double* J = v.prior_out + ((size_t)b * v.B + w) * PRIOR_OUT + 15;
        for (int r = 0; r < 15; r++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 15; c++) s = fma(J[r * 15 + c], pj[c], s);
            q.u_pri[(size_t)w * 15 + r] = s;
        }
    }
    if (k == lo) return;
    double pi[15], u[15];
#pragma unroll
    for (int c = 0; c < 15; c++) { pi[c] = p[(size_t)(gk - 1) * 15 + c]; u[c] = 0.0; }
    const double* __restrict__ tile = v.imu_j + ((size_t)b * (size_t)(v.G >> JT_LOG) + (size_t)(gk >> JT_LOG)) * JT_STRIDE + (gk & (JT - 1)) * 2;
#pragma unroll
    for (int col = 0; col < 30; col++) {
#pragma unroll
        for (int row = 0; row < 15; row++) {
            const int e = JM.idx[row * 30 + col];            // compile-time once unrolled
            if (e < 0) continue;
            const int side = jcol_is_j(col) ? 1 : 0;
            const double val = tile[(size_t)((side ? JS_PI : 0) + (e >> 1)) * (JT * 2) + (e & 1)];
            u[row] = fma(val, side ? pj[col_comp(col)] : pi[col_comp(col)], u[row]);
        }
    }
#pragma unroll
    for (int r = 0; r < 15; r++) q.u_imu[(size_t)r * v.G + gk] = u[r];
    const int a = v.btw_a[gk];
    if (a >= lo && a < k) {
        const double* __restrict__ fb = v.btw_out + ((size_t)b * (size_t)(v.G >> 6) + (size_t)(gk >> 6)) * BTW_OUT * TILE + (gk & 63);
        const double* pa = p + ((size_t)w * v.M + a) * 15;
        double pa6[6];
#pragma unroll
        for (int c = 0; c < 6; c++) pa6[c] = pa[c];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) s = fma(fb[(size_t)(6 + r * 6 + c) * TILE], pa6[c], fma(fb[(size_t)(42 + r * 6 + c) * TILE], pj[c], s));
            q.u_btw[(size_t)r * v.G + gk] = s;
        }
    }
}

// one side (columns of keyframe `side_j`) of J^T u of the IMU factor in slot g: o[c] += sum_row J[row][imu_col(side, c)] u[row]
template <int SIDE, bool GRAD>
VF_DI void jt_side(const View& v, const Refine& q, int b, long g, double (&o)[15]) {
    double u[15];
    const double* __restrict__ res = v.imu_r + ((size_t)b * (size_t)(v.G >> 6) + (size_t)(g >> 6)) * IMU_R * TILE + (g & 63);
#pragma unroll
    for (int r = 0; r < 15; r++) u[r] = GRAD ? res[(size_t)r * TILE] : q.u_imu[(size_t)r * v.G + g];
    const double* __restrict__ tile = v.imu_j + ((size_t)b * (size_t)(v.G >> JT_LOG) + (size_t)(g >> JT_LOG)) * JT_STRIDE + (g & (JT - 1)) * 2;
#pragma unroll
    for (int c = 0; c < 15; c++) {
#pragma unroll
        for (int row = 0; row < 15; row++) {
            const int e = JM.idx[row * 30 + imu_col(SIDE, c)];
            if (e < 0) continue;
            o[c] = fma(tile[(size_t)((SIDE ? JS_PI : 0) + (e >> 1)) * (JT * 2) + (e & 1)], u[row], o[c]);
        }
    }
}

// out = J^T u + lambda p, one lane per keyframe (owner computes: the sums of a keyframe are formed in one fixed order),
// plus the marginal prior's information times p (it is kept in information form: 27 x 27, on [lo: 15][lo+1: pose][lo+2: pose]).
// GRAD: out = J^T r, the gradient at the current linearisation, from the residuals the linearisation kernels left (a
// time-sharded rank assembles g for its own rows only, but holds every residual and every Jacobian).
template <bool GRAD>
__global__ void __launch_bounds__(256) k_jtu(View v, Refine q, const double* __restrict__ p, double* __restrict__ out) {
    const long gk = (long)blockIdx.x * 256 + threadIdx.x;
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    const int lo = v.lo[w], hi = v.hi[w], b = v.sel[w];
    if (k < lo || k >= hi || refine_off(v, q, w)) return;
    const double lam = v.lambda[w];
    double o[15];
#pragma unroll
    for (int c = 0; c < 15; c++) o[c] = GRAD ? 0.0 : lam * p[(size_t)gk * 15 + c];
    if (k > lo) jt_side<1, GRAD>(v, q, b, gk, o);
    if (k + 1 < hi) jt_side<0, GRAD>(v, q, b, gk + 1, o);
    const size_t tiles = (size_t)(v.G >> 6);
    if (k > lo) {
        const int a = v.btw_a[gk];
        if (a >= lo && a < k) {
            const double* __restrict__ fb = v.btw_out + ((size_t)b * tiles + (size_t)(gk >> 6)) * BTW_OUT * TILE + (gk & 63);
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const double ur = GRAD ? fb[(size_t)r * TILE] : q.u_btw[(size_t)r * v.G + gk];
#pragma unroll
                for (int c = 0; c < 6; c++) o[c] = fma(fb[(size_t)(42 + r * 6 + c) * TILE], ur, o[c]);
            }
        }
    }
    for (int d = 1; d <= 3; d++) {
        const long g2 = gk + d;
        if (k + d >= hi || v.btw_a[g2] != k) continue;
        const double* __restrict__ fb = v.btw_out + ((size_t)b * tiles + (size_t)(g2 >> 6)) * BTW_OUT * TILE + (g2 & 63);
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const double ur = GRAD ? fb[(size_t)r * TILE] : q.u_btw[(size_t)r * v.G + g2];
#pragma unroll
            for (int c = 0; c < 6; c++) o[c] = fma(fb[(size_t)(6 + r * 6 + c) * TILE], ur, o[c]);
        }
    }
    if (v.prior_k[w] == k) {
        const double* J = v.prior_out + ((size_t)b * v.B + w) * PRIOR_OUT + 15;
        for (int r = 0; r < 15; r++) {
            const double ur = GRAD ? J[r - 15] : q.u_pri[(size_t)w * 15 + r];
#pragma unroll
            for (int c = 0; c < 15; c++) o[c] = fma(J[r * 15 + c], ur, o[c]);
        }
    }
    if (v.mp_on[w] && hi - lo >= 3 && k - lo < 3) {
        const int j = k - lo, r0 = j == 0 ? 0 : 15 + 6 * (j - 1), nr = j == 0 ? 15 : 6;
        const double* L = v.mp_L + (size_t)w * 729;
        const double* p0 = p + ((size_t)w * v.M + lo) * 15;
        if (GRAD) {          // the marginal prior's gradient L d + eta, as k_linearize_prior left it
            const double* gm = v.mp_out + ((size_t)b * v.B + w) * 28;
            for (int c = 0; c < nr; c++) o[c] += gm[r0 + c];
        } else
        for (int i = 0; i < 27; i++) {
            const double pv = i < 15 ? p0[i] : (i < 21 ? p0[15 + (i - 15)] : p0[30 + (i - 21)]);
            for (int c = 0; c < nr; c++) o[c] = fma(L[(r0 + c) * 27 + i], pv, o[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 15; c++) out[(size_t)gk * 15 + c] = o[c];
}

// ---- far between factors (View::x_*, xl_*; vf_far.hpp): six more rows of J per slot.  One workgroup per window, after k_jtu:
// out += U^T (U p) (GRAD: out += U^T r), slot by slot (two slots may touch the same keyframe) and, within a slot, one lane
// per column, the lane of the first of several columns that land on one (keyframe, dof) adding them all.
#include "vf_far.hpp"
template <bool GRAD>
__global__ void __launch_bounds__(64) k_far_apply(View v, Refine q, const double* __restrict__ p, double* __restrict__ out) {
    const int w = blockIdx.x, lane = threadIdx.x;
    if (v.hi[w] - v.lo[w] <= 0 || refine_off(v, q, w)) return;
    const int b = v.sel[w];
    __shared__ double u[6];
    __shared__ int kb_l[MAX_EXTRA_BIG];                // far ends of the linear far factor (every linear slot's columns 27.. lie on them)
    if (lane < v.x_max) kb_l[lane] = v.xl_b[w * v.x_max + lane];
    __syncthreads();
    for (int s = 0; s < v.x_max; s++) {
        const FarRef f = far_ref(v, w, s);
        if (f.kind < 0) continue;                    // (the same for every lane)
        const int nc = far_cols(f);
        if (lane < 6) {
            double acc = GRAD ? far_res(v, w, f, b, lane) : 0.0;
            if (!GRAD)
                for (int c = 0; c < nc; c++) {
                    int k, d;
                    far_col(v, w, f, c, k, d);
                    acc = fma(far_jac(v, w, f, b, lane, c), p[((size_t)w * v.M + k) * 15 + d], acc);
                }
            u[lane] = acc;
        }
        __syncthreads();
        for (int c0 = 0; c0 < nc; c0 += 64) {
            const int c = c0 + lane;
            if (c < nc) {
                int k, d;
                far_col(v, w, f, c, k, d);
                double acc = 0.0;
                for (int r = 0; r < 6; r++) acc = fma(far_jac(v, w, f, b, r, c), u[r], acc);
                bool first = true;
                if (f.kind == 1 && c >= 27)
                    for (int c2 = 27 + d; c2 < nc; c2 += 6) {
                        if (c2 == c || kb_l[(c2 - 27) / 6] != k) continue;
                        if (c2 < c) { first = false; break; }
                        for (int r = 0; r < 6; r++) acc = fma(far_jac(v, w, f, b, r, c2), u[r], acc);
                    }
                if (first) out[((size_t)w * v.M + k) * 15 + d] += acc;
            }
        }
        __syncthreads();
    }
}

// ---- whole-vector steps at the ends of a refinement (PCG_NB workgroups per window, as the steps of a correction below)
constexpr int PCG_NB = 64, PCG_NT = 256;
// x := delta (the plain normal-equation solution the engine's solve has just left); a window whose factorisation failed
// takes no part
__global__ void __launch_bounds__(PCG_NT) k_pcg_begin(View v, Refine q) {
    const int w = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
    const int lo = v.lo[w], hi = v.hi[w], n = (hi - lo) * 15;
    if (bx == 0 && tid == 0) { q.stop[w] = (solve_failed(v, w) || hi - lo <= 0 || (v.stop_on && v.done[w])) ? 1 : 0; q.rz[w] = -1.0; q.rz0[w] = 0.0; q.iters[w] = 0; }
    if (n <= 0) return;
    const size_t o = ((size_t)w * v.M + lo) * 15;
    const int per = (n + PCG_NB - 1) / PCG_NB, e0 = bx * per, e1 = min(n, e0 + per);
    for (int e = e0 + tid; e < e1; e += PCG_NT) q.x[o + e] = v.delta[o + e];
}
// nres := g + A x  (= minus the residual of the normal equations at x, with A x evaluated through J; g = J^T r is in q.p,
// formed by k_jtu<true>: the same on every rank of a time-sharded window, whose K3 assembles only the rank's own rows)
__global__ void __launch_bounds__(PCG_NT) k_pcg_residual(View v, Refine q) {
    const int w = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
    if (refine_off(v, q, w)) return;
    const int lo = v.lo[w], hi = v.hi[w], n = (hi - lo) * 15;
    const size_t o = ((size_t)w * v.M + lo) * 15;
    const int per = (n + PCG_NB - 1) / PCG_NB, e0 = bx * per, e1 = min(n, e0 + per);
    for (int e = e0 + tid; e < e1; e += PCG_NT) q.nres[o + e] = q.p[o + e] + q.Ap[o + e];
}
// ---- the two dot products and the two vector updates of a correction.  A window can be 10^4 keyframes long (150 000
// entries): each step is three small launches -- partial sums by PCG_NB workgroups per window, one lane per window that adds
// the partials in a fixed order and derives the step's scalars, PCG_NB workgroups per window applying them -- instead of one
// 1024-thread workgroup walking the whole window (measured on the 10 000-pose window: 143 + 94 us per correction before,
// against 600 us for the band solve the correction is made for).  Sums are formed in one fixed order whatever the rank.
// mode 0: res . z = -nres . z ;  mode 1: p . A p
__global__ void __launch_bounds__(PCG_NT) k_pcg_dot(View v, Refine q, int mode) {
    __shared__ double red[PCG_NT];
    const int w = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
    if (refine_off(v, q, w)) return;
    const int lo = v.lo[w], hi = v.hi[w], n = (hi - lo) * 15;
    const size_t o = ((size_t)w * v.M + lo) * 15;
    const int per = (n + PCG_NB - 1) / PCG_NB, e0 = bx * per, e1 = min(n, e0 + per);
    const double* a = mode == 0 ? q.nres : q.p;
    const double* b = mode == 0 ? q.z : q.Ap;
    double s = 0.0;
    for (int e = e0 + tid; e < e1; e += PCG_NT) s = fma(a[o + e], b[o + e], s);
    red[tid] = s;
    __syncthreads();
    for (int st = PCG_NT / 2; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) q.part[(size_t)w * PCG_NB + bx] = mode == 0 ? -red[0] : red[0];
}
// one lane per window: the step's scalars.  mode 0 (after a correction solve z = M^-1 res): rz' = res . z, the stopping rule,
// beta = rz' / rz (coef; first direction: p := z);  mode 1: alpha = rz / (p . A p) (coef)
__global__ void __launch_bounds__(64) k_pcg_scalar(View v, Refine q, int mode, double rel_stop) {
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w >= v.B || refine_off(v, q, w)) return;
    double s = 0.0;
    for (int i = 0; i < PCG_NB; i++) s += q.part[(size_t)w * PCG_NB + i];
    if (mode == 0) {
        const double rz = q.rz[w], rz0 = q.rz0[w];
        const bool first = rz < 0.0;
        // stop: the preconditioned residual has lost rel_stop^2 of its first value, or is no longer positive (rounding floor,
        // or a correction solve that failed)
        if (!(s > 0.0) || (!first && s <= rel_stop * rel_stop * rz0)) { q.stop[w] = 1; return; }
        q.coef[w] = first ? 0.0 : s / rz;
        q.first[w] = first ? 1 : 0;
        q.rz[w] = s;
        if (first) q.rz0[w] = s;
    } else {
        if (!(s > 0.0)) { q.stop[w] = 1; return; }
        q.coef[w] = q.rz[w] / s;
        q.iters[w] += 1;
    }
}
// mode 0: p := z + beta p (first: p := z);  mode 1: x += alpha p, nres += alpha A p
__global__ void __launch_bounds__(PCG_NT) k_pcg_axpy(View v, Refine q, int mode) {
    const int w = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
    if (refine_off(v, q, w)) return;
    const int lo = v.lo[w], hi = v.hi[w], n = (hi - lo) * 15;
    const size_t o = ((size_t)w * v.M + lo) * 15;
    const int per = (n + PCG_NB - 1) / PCG_NB, e0 = bx * per, e1 = min(n, e0 + per);
    const double c = q.coef[w];
    if (mode == 0) {
        const bool first = q.first[w] != 0;
        for (int e = e0 + tid; e < e1; e += PCG_NT) q.p[o + e] = first ? q.z[o + e] : fma(c, q.p[o + e], q.z[o + e]);
    } else {
        for (int e = e0 + tid; e < e1; e += PCG_NT) {
            q.x[o + e] = fma(c, q.p[o + e], q.x[o + e]);
            q.nres[o + e] = fma(c, q.Ap[o + e], q.nres[o + e]);
        }
    }
}
// delta := x
__global__ void __launch_bounds__(PCG_NT) k_pcg_end(View v, Refine q) {
    const int w = blockIdx.y, bx = blockIdx.x, tid = threadIdx.x;
    const int lo = v.lo[w], hi = v.hi[w], n = (hi - lo) * 15;
    if (n <= 0 || solve_failed(v, w) || (v.stop_on && v.done[w])) return;
    const size_t o = ((size_t)w * v.M + lo) * 15;
    const int per = (n + PCG_NB - 1) / PCG_NB, e0 = bx * per, e1 = min(n, e0 + per);
    for (int e = e0 + tid; e < e1; e += PCG_NT) v.delta[o + e] = q.x[o + e];
}

static inline unsigned nblk_(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }
void launch_refine_apply(const View& v, const Refine& q, const double* p, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_jv, dim3(nblk_(v.G, 256)), dim3(256), 0, s, v, q, p);
    hipLaunchKernelGGL(k_jtu<false>, dim3(nblk_(v.G, 256)), dim3(256), 0, s, v, q, p, out);
    if (v.x_max > 0) hipLaunchKernelGGL(k_far_apply<false>, dim3(v.B), dim3(64), 0, s, v, q, p, out);
}
void launch_refine_begin(const View& v, const Refine& q, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_begin, dim3(PCG_NB, (unsigned)v.B), dim3(PCG_NT), 0, s, v, q);
    launch_refine_apply(v, q, q.x, q.Ap, s);
    hipLaunchKernelGGL(k_jtu<true>, dim3(nblk_(v.G, 256)), dim3(256), 0, s, v, q, q.x, q.p);     // q.p := J^T r (p has no direction yet)
    if (v.x_max > 0) hipLaunchKernelGGL(k_far_apply<true>, dim3(v.B), dim3(64), 0, s, v, q, q.x, q.p);
    hipLaunchKernelGGL(k_pcg_residual, dim3(PCG_NB, (unsigned)v.B), dim3(PCG_NT), 0, s, v, q);
}
void launch_refine_step(const View& v, const Refine& q, double rel_stop, hipStream_t s) {
    const dim3 grid(PCG_NB, (unsigned)v.B);
    hipLaunchKernelGGL(k_pcg_dot, grid, dim3(PCG_NT), 0, s, v, q, 0);
    hipLaunchKernelGGL(k_pcg_scalar, dim3(nblk_(v.B, 64)), dim3(64), 0, s, v, q, 0, rel_stop);
    hipLaunchKernelGGL(k_pcg_axpy, grid, dim3(PCG_NT), 0, s, v, q, 0);
    launch_refine_apply(v, q, q.p, q.Ap, s);
    hipLaunchKernelGGL(k_pcg_dot, grid, dim3(PCG_NT), 0, s, v, q, 1);
    hipLaunchKernelGGL(k_pcg_scalar, dim3(nblk_(v.B, 64)), dim3(64), 0, s, v, q, 1, rel_stop);
    hipLaunchKernelGGL(k_pcg_axpy, grid, dim3(PCG_NT), 0, s, v, q, 1);
}
void launch_refine_end(const View& v, const Refine& q, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_end, dim3(PCG_NB, (unsigned)v.B), dim3(PCG_NT), 0, s, v, q);
}

}  // namespace vf
