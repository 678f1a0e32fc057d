// vf_kernels.hip -- gfx950 kernels of the smoother hot path.
//
//   K1 k_linearize_imu      CombinedImuFactor residual + whitened 15x30 Jacobian
//                           (replaces the linearisation ISAM2::update performs for the factor
//                           built at gtsam_fusion/src/gtsam_fusion/IMUManager.cpp:68-73)
//   K2 k_linearize_between  BetweenFactor<Pose3> residual + whitened 6x6 Jacobians
//                           (factor built at GraphManager.cpp:86)
//   K2b k_linearize_prior   the three priors of GraphManager.cpp:27-35
//   K3 k_assemble           block-banded J^T J / J^T r, owner-computes per keyframe
//   K0 k_preintegrate       combined-IMU preintegration (IMUManager.cpp:42-64 + GTSAM's PIM)
//   K4 k_band_solve(_tw)    (H + lambda I) delta = -g, block-banded Cholesky, one wave per window
//                           (_tw: two waves from both ends)
//   K4p k_chunk_forward, k_sep_solve, k_chunk_rhs, k_chunk_back
//                           the same solve partitioned into chunks joined by 27-dof separators
//                           (one-window latency; the per-GPU piece of a time-sharded window)
//   K-marg k_marginalize    fixed-lag marginalisation of the oldest keyframe into a dense prior
//   K5 k_retract, k_decide  x (+) delta, cost reduction, LM accept/reject
//   a2 k_predict, k_slide   PreintegrationBase::predict initial values (GraphManager.cpp:152-160)
//
// Mapping: one lane per factor / keyframe for K1-K3,K5 (HBM-bound, coalesced AoSoA tiles of 64:
// lane l of a wave reads word l of a 512-byte field row), one 64-lane wave per window or per
// chunk for K4 / K4p (panel factorisation in registers with v_readlane / DPP broadcasts,
// trailing window in LDS, Schur / spike products on v_mfma_f64_16x16x4).
#include "vf_kernels.hpp"
#include "vf_jstream.hpp"
#include "vf_math.hpp"
#include <cstdlib>

namespace vf {

#define XS(buf, c, gk) v.x[((size_t)(buf) * 16 + (c)) * (size_t)v.G + (size_t)(gk)]

// optional LM termination: a window that has converged takes no part in the remaining trials of this solve
VF_DI bool window_done(const View& v, int w) { return v.stop_on && v.done[w]; }
// hybrid K4: which of the two forms of this launch does the work (see View::gate)
VF_DI bool gated_off(const View& v) {
    if (!v.gate) return false;
    const int na = *v.n_active;
    return v.gate == 1 ? na <= v.gate_T : na > v.gate_T;
}

// one-wave sweeps: workgroup `slot` -> window (identity unless the launch carries the compacted list of active windows)
VF_DI int sweep_window(const View& v, int slot) {
    if (!v.act) return slot;
    return slot < *v.n_active ? v.act[slot] : -1;
}

// Time-sharded windows: the window-local keyframe range [klo, khi) and the chunk range [c0, c1) rank sh_r owns.
// A chunk owns its interior keyframes and the separator that follows it; a factor belongs to its later keyframe.
VF_DI void own_chunks(const View& v, int Pe, int& c0, int& c1) {
    if (v.sh_G <= 1) { c0 = 0; c1 = Pe; return; }
    c0 = (int)((long)v.sh_r * Pe / v.sh_G);
    c1 = (int)((long)(v.sh_r + 1) * Pe / v.sh_G);
}
VF_DI void own_range(const View& v, int w, int& klo, int& khi) {
    const int n = v.hi[w] - v.lo[w];
    klo = 0;
    khi = n;
    if (v.sh_G <= 1 || n <= 0) return;
    const int Pe = chunk_count(n, v.P, v.P_fit);
    int c0, c1;
    own_chunks(v, Pe, c0, c1);
    klo = c0 < Pe ? chunk_geom(n, Pe, c0).i0 : n;
    khi = c1 < Pe ? chunk_geom(n, Pe, c1).i0 : n;
}
// A rank's last chunk sweeps two rows beyond its own keyframes (the pose rows of the two keyframes after its cut
// keyframe), so it assembles the rows [klo, khi + 2) of H, and needs the linearisation of the factor slots that
// feed them: 3 more (a row of H collects factors up to 3 keyframes ahead).
VF_DI bool shard_skips_factor(const View& v, int w, int k) {
    if (v.sh_G <= 1 || v.sh_all_jac) return false;
    int klo, khi;
    own_range(v, w, klo, khi);
    const int kk = k - v.lo[w];
    return kk < klo || kk >= khi + 5;
}

struct State {
    Q4 q;
    V3 t, vel, ba, bg;
};

VF_DI State load_state(const View& v, int buf, long gk) {
    State s;
    s.q = q4(XS(buf, 0, gk), XS(buf, 1, gk), XS(buf, 2, gk), XS(buf, 3, gk));
    s.t = v3(XS(buf, 4, gk), XS(buf, 5, gk), XS(buf, 6, gk));
    s.vel = v3(XS(buf, 7, gk), XS(buf, 8, gk), XS(buf, 9, gk));
    s.ba = v3(XS(buf, 10, gk), XS(buf, 11, gk), XS(buf, 12, gk));
    s.bg = v3(XS(buf, 13, gk), XS(buf, 14, gk), XS(buf, 15, gk));
    return s;
}
VF_DI void store_state(const View& v, int buf, long gk, const State& s) {
    XS(buf, 0, gk) = s.q.w; XS(buf, 1, gk) = s.q.x; XS(buf, 2, gk) = s.q.y; XS(buf, 3, gk) = s.q.z;
    XS(buf, 4, gk) = s.t.x; XS(buf, 5, gk) = s.t.y; XS(buf, 6, gk) = s.t.z;
    XS(buf, 7, gk) = s.vel.x; XS(buf, 8, gk) = s.vel.y; XS(buf, 9, gk) = s.vel.z;
    XS(buf, 10, gk) = s.ba.x; XS(buf, 11, gk) = s.ba.y; XS(buf, 12, gk) = s.ba.z;
    XS(buf, 13, gk) = s.bg.x; XS(buf, 14, gk) = s.bg.y; XS(buf, 15, gk) = s.bg.z;
}

// packed upper-triangular index helpers (row-major, row r holds columns r..n-1)
__host__ __device__ constexpr int off15(int r) { return r * 15 - r * (r - 1) / 2; }
__host__ __device__ constexpr int idx9(int a, int b) { return a * 9 - a * (a - 1) / 2 + (b - a); }
__host__ __device__ constexpr int off6(int r) { return r * 6 - r * (r - 1) / 2; }

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));
template <int N> struct IC { static constexpr int value = N; };     // compile-time integer as a function argument

VF_DI double col3(const M3& A, int r, int c) { return A.a[r * 3 + c]; }

// o[a] = sum_{b >= a, LO <= b < HI} R11(a,b) u[b]   (upper-triangular 9x9 times sparse column)
template <int LO, int HI>
VF_DI void white9(const double (&R)[45], const double (&u)[9], double (&o)[9]) {
#pragma unroll
    for (int a = 0; a < 9; a++) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < 9; b++)
            if (b >= a && b >= LO && b < HI) s = fma(R[idx9(a, b)], u[b], s);
        o[a] = s;
    }
}

// ------------------------------------------------------------------------------------ K0
// IMU preintegration, one 256-thread workgroup per factor: PreintegratedCombinedMeasurements::
// integrateMeasurement over the factor's steps (mean, bias Jacobians, 15x15 covariance; GTSAM 4.0.x tangent
// form), then the noise model R = chol_upper(cov^-1) of the CombinedImuFactor built at
// gtsam_fusion/src/gtsam_fusion/IMUManager.cpp:68-73.  Runs once per factor (not per LM iteration).
// F, P and F P live in LDS, thread (i, j) owns entry (i, j) of the 15x15 products; the small 3x3 quantities of a
// step are computed by every thread.  (The first version ran one LANE per factor with the 15x15 work in per-lane
// scratch arrays: 2-4 ms whatever the batch, i.e. most of a GraphManager::solve, which preintegrates one factor.)
// Every entry is accumulated in the same order as a plain triple loop would, so the result does not depend on
// the mapping.
// TAIL = false: factors [g0, g0 + n) of consecutive slots, bias estimates handed in (vf_engine_preintegrate).
// TAIL = true (vf_engine_ingest_tail, the ingest half of a fixed-lag update for ALL windows in one launch): block w
// preintegrates the factor that ends at window w's NEXT keyframe, slot hi[w], with the window's CURRENT bias estimate -- the
// bias of its last keyframe in the current buffer, what GraphManager::reserveNode passes as getBias()
// (GraphManager.cpp:59) -- and stages the between record that ends at the same keyframe (GraphManager.cpp:83-88).
template <bool TAIL>
__global__ void __launch_bounds__(256) k_preintegrate_t(View v, long g0, int n, const int* __restrict__ off,
                                                       const double* __restrict__ steps,
                                                       const double* __restrict__ bhat6, const int* __restrict__ tail_a,
                                                       const double* __restrict__ tail_btw, ImuCov prm, int* status) {
    const int f = blockIdx.x, tid = threadIdx.x;
    if (f >= n) return;
    __shared__ double sF[225], sP[225], sT[225], sH[54], sHn[54];
    __shared__ int s_ok;
    const int i = tid / 15, j = tid - i * 15;          // entry of the 15x15 matrices (tid < 225)
    const int hi_ = tid / 6, hj = tid - hi_ * 6;       // entry of the 9x6 bias Jacobian (tid < 54)
    long gk = g0 + f;
    double bh[6];
    if (TAIL) {
        const int hi_w = v.hi[f];
        if (hi_w >= v.M || hi_w <= v.lo[f]) {          // no free slot / empty window: reported, nothing written
            if (tid == 0) atomicOr(status, 2);
            return;
        }
        gk = (long)f * v.M + hi_w;
        const int b = v.sel[f];
#pragma unroll
        for (int c = 0; c < 6; c++) bh[c] = XS(b, 10 + c, gk - 1);
    } else {
#pragma unroll
        for (int c = 0; c < 6; c++) bh[c] = bhat6[(size_t)f * 6 + c];
    }
    const V3 bacc = v3(bh[0], bh[1], bh[2]), bgyr = v3(bh[3], bh[4], bh[5]);
    V3 th = v3(0, 0, 0), pos = v3(0, 0, 0), vel = v3(0, 0, 0);
    double dtij = 0.0;
    if (tid < 225) sP[tid] = 0.0;
    if (tid < 54) sH[tid] = 0.0;
    if (tid == 0) s_ok = 1;
    __syncthreads();
    for (int s = off[f]; s < off[f + 1]; s++) {
        const double* st = steps + (size_t)s * 7;
        const double dt = st[0], dt22 = 0.5 * dt * dt;
        const V3 acc = v3(st[1], st[2], st[3]) - bacc, om = v3(st[4], st[5], st[6]) - bgyr;
        const M3 Jr = so3_jr(th), invD = so3_jr_inv(th);
        const V3 wt = mul(invD, om);
        const M3 R = qrot(qexp(th));
        const V3 anav = mul(R, acc);
        const M3 wH = mul(invD, so3_jr_apply_dtheta(th, wt));   // -w_tangent_H_theta
        const M3 aH = mul(mulSkew(R, neg(acc)), Jr);            // a_nav_H_theta
        // F = [[A, Fb], [0, I]]
        if (tid < 225) {
            double x = i == j ? 1.0 : 0.0;
            if (i < 3 && j < 3) x -= wH.a[i * 3 + j] * dt;
            if (i >= 3 && i < 6 && j < 3) x = aH.a[(i - 3) * 3 + j] * dt22;
            if (i >= 6 && i < 9 && j < 3) x = aH.a[(i - 6) * 3 + j] * dt;
            if (i < 3 && j >= 12) x = -invD.a[i * 3 + j - 12] * dt;                 // theta_H_biasOmega = -C.top
            if (i >= 6 && i < 9 && j >= 9 && j < 12) x = -R.a[(i - 6) * 3 + j - 9] * dt;   // vel_H_biasAcc = -B.bottom
            if (i >= 3 && i < 6 && j == i + 3) x = dt;
            sF[tid] = x;
        }
        __syncthreads();
        // bias Jacobians: H <- A H - [B | C]
        if (tid < 54) {
            double a = 0.0;
            for (int l = 0; l < 9; l++) a = fma(sF[hi_ * 15 + l], sH[l * 6 + hj], a);
            if (hi_ >= 3 && hi_ < 6 && hj < 3) a -= R.a[(hi_ - 3) * 3 + hj] * dt22;
            if (hi_ >= 6 && hj < 3) a -= R.a[(hi_ - 6) * 3 + hj] * dt;
            if (hi_ < 3 && hj >= 3) a -= invD.a[hi_ * 3 + hj - 3] * dt;
            sHn[tid] = a;
        }
        // covariance: P <- F P F^T + G Q G^T
        if (tid < 225) {
            double a = 0.0;
            for (int l = 0; l < 15; l++) a = fma(sF[i * 15 + l], sP[l * 15 + j], a);
            sT[tid] = a;
        }
        __syncthreads();
        if (tid < 54) sH[tid] = sHn[tid];
        if (tid < 225) {
            double a = 0.0;
            for (int l = 0; l < 15; l++) a = fma(sT[i * 15 + l], sF[j * 15 + l], a);
            const double sv = (prm.acc + prm.bias_int) * dt, sr = (prm.gyro + prm.bias_int) * dt;
            if (i >= 6 && i < 9 && j >= 6 && j < 9) {           // (1/dt) vHb (aCov+int) vHb^T
                const M3 RRt = mulBT(R, R);
                a += sv * RRt.a[(i - 6) * 3 + j - 6];
            }
            if (i < 3 && j < 3) {                               // (1/dt) tHb (wCov+int) tHb^T
                const M3 DDt = mulBT(invD, invD);
                a += sr * DDt.a[i * 3 + j];
            }
            if (i == j && i >= 3 && i < 6) a += dt * prm.integration;
            if (i == j && i >= 9 && i < 12) a += dt * prm.bias_acc;
            if (i == j && i >= 12) a += dt * prm.bias_omega;
            sP[tid] = a;
        }
        // mean
        th = th + dt * wt;
        pos = pos + dt * vel + dt22 * anav;
        vel = vel + dt * anav;
        dtij += dt;
        __syncthreads();
    }
    // R upper with R^T R = P^-1: reverse Cholesky P = U U^T (U upper, in sF), R = U^-1 (in sT)
    if (tid < 225) { sF[tid] = 0.0; sT[tid] = 0.0; }
    __syncthreads();
    for (int c = 14; c >= 0; c--) {
        if (tid == 0) {
            double d = sP[c * 15 + c];
            for (int l = c + 1; l < 15; l++) d = fma(-sF[c * 15 + l], sF[c * 15 + l], d);
            if (!(d > 0.0)) { s_ok = 0; d = 1.0; }
            sF[c * 15 + c] = sqrt(d);
        }
        __syncthreads();
        if (tid < c) {                                          // row tid of column c
            const double ujj = sF[c * 15 + c];
            double a = 0.5 * (sP[tid * 15 + c] + sP[c * 15 + tid]);
            for (int l = c + 1; l < 15; l++) a = fma(-sF[tid * 15 + l], sF[c * 15 + l], a);
            sF[tid * 15 + c] = a / ujj;
        }
        __syncthreads();
    }
    if (tid < 15) {                                             // column tid of U^-1, bottom up
        const int c = tid;
        sT[c * 15 + c] = 1.0 / sF[c * 15 + c];
        for (int r = c - 1; r >= 0; r--) {
            double a = 0.0;
            for (int l = r + 1; l <= c; l++) a = fma(sF[r * 15 + l], sT[l * 15 + c], a);
            sT[r * 15 + c] = -a / sF[r * 15 + r];
        }
    }
    __syncthreads();
    double* out = v.imu_in + (size_t)(gk >> 6) * IMU_IN * TILE + (gk & 63);
    if (TAIL) {
        if (tid >= 224 && tid < 224 + BTW_IN)
            v.btw_in[((size_t)(gk >> 6) * BTW_IN + (tid - 224)) * TILE + (gk & 63)] = tail_btw[(size_t)f * BTW_IN + tid - 224];
        if (tid == 255) v.btw_a[gk] = tail_a[f];
    }
    if (tid < IMU_IN) {
        double x;
        if (tid == 0) x = dtij;
        else if (tid < 4) x = tid == 1 ? th.x : (tid == 2 ? th.y : th.z);
        else if (tid < 7) x = tid == 4 ? pos.x : (tid == 5 ? pos.y : pos.z);
        else if (tid < 10) x = tid == 7 ? vel.x : (tid == 8 ? vel.y : vel.z);
        else if (tid < 16) x = tid == 10 ? bh[0] : (tid == 11 ? bh[1] : (tid == 12 ? bh[2] : (tid == 13 ? bh[3] : (tid == 14 ? bh[4] : bh[5]))));
        else if (tid < 70) x = sH[tid - 16];
        else {                                                  // packed upper triangle, row-major
            int o = tid - 70, r = 0;
            while (o >= 15 - r) { o -= 15 - r; r++; }
            x = sT[r * 15 + r + o];
        }
        out[(size_t)tid * TILE] = x;
    }
    if (tid == 0 && (!s_ok || off[f + 1] == off[f])) atomicOr(status, 1);
}

// ------------------------------------------------------------------------------------ K1
// Algorithmic traffic per factor: 222 doubles in (2 states x 16, record 190), 465 out.
#ifndef VF_K1_BLOCK
#define VF_K1_BLOCK 256
#endif
#ifndef VF_K1_WAVES
#define VF_K1_WAVES 1
#endif
VF_DI const double* jtile_ptr(const double* base, long gk) { return base + (size_t)(gk >> JT_LOG) * JT_STRIDE; }
// entry (row, col) of the Jacobian of the factor in slot gk of a J stream buffer (0 for a structural zero)
VF_DI double jstream_entry(const double* jbuf, long gk, int row, int col) {
    const int e = JMD.idx[row * 30 + col];
    if (e < 0) return 0.0;
    const int pair = (jcol_is_j(col) ? JS_PI : 0) + (e >> 1);
    return jbuf[(size_t)(gk >> JT_LOG) * JT_STRIDE + ((size_t)pair * JT + (gk & (JT - 1))) * 2 + (e & 1)];
}

// Where a factor's (r | J) goes.  K1: r into the AoSoA residual array, J into the factor's slot of its J-stream tile as
// 16-byte non-temporal stores of two consecutive entries (write-once streams far beyond L2 / MALL: +15 % measured for
// non-temporal).  (A copy of the i-side pairs of a tile's first factor into the tile in front of it, so that K3 would
// find its halo factor in its own block, cost K1 0.48 ms: 16-byte partial-line writes.  K3 reads the halo where it is.)
// `jac` = false (time-sharded windows, factors this rank does not assemble): residual only.
struct HbmSink {
    // The 9x6 bias Jacobians of the record are used twice (bias-corrected delta, bias columns of J) and the second read
    // misses L2 (18 % excess fetch traffic in the PMC counters).  Keeping them in registers (keep_h = true: 352 -> 406)
    // removes that traffic (2.15 -> 1.71 GB read per launch) and makes K1 SLOWER, 0.80 -> 0.855 ms: with one wave per SIMD
    // the kernel is paced by how early its first compute can start, not by its byte count.  Measured, left off.
    static constexpr bool keep_h = false;
    double* out_r;      // imu_r row of this factor (+ a * TILE)
    double* slot;       // imu_j: tile base + 2 * slot
    bool jac;
    double pend[2];
    VF_DI void r(int a, double x) { __builtin_nontemporal_store(x, out_r + (size_t)a * TILE); }
    VF_DI void pair(int side, int p, double x0, double x1) {
        if (!jac) return;
        d2_t t;
        t.x = x0;
        t.y = x1;
        __builtin_nontemporal_store(t, (d2_t*)(slot + (size_t)((side ? JS_PI : 0) + p) * (JT * 2)));
    }
    VF_DI void j(int row, int col, double x) {
        const int e = JM.idx[row * 30 + col];          // compile-time constants once the core's loops are unrolled
        const int side = jcol_is_j(col) ? 1 : 0;
        if (e & 1) pair(side, e >> 1, pend[side], x);
        else if (e == JM.n[side] - 1) pair(side, e >> 1, x, 0.0);   // odd count: the last entry goes with a zero
        else pend[side] = x;
    }
};
// LDS image of one IMU linearisation = its words of the J stream as they are ([0, 2 JS_PAIRS): i-side pairs then j-side
// pairs), then r (15) and a zero cell the MFMA operand maps point at for structural zeros and padding
constexpr int LJ_R = 2 * JS_PAIRS, LJ_ZERO = LJ_R + 15;
constexpr int LJS = 310;        // LDS stride of one factor (even: the staging writes are 16-byte; 620 dwords = 12 mod 32 banks: the 8 slots of a pair hit 8 different bank groups)
// CombinedImuFactor: residual and whitened 15x30 Jacobian of the factor in slot gk, at the states of buffer b.
template <class Sink>
__device__ __forceinline__ void linearize_imu_core(const View& v, const int b, const long gk, Sink& sink) {
    const double* __restrict__ in = v.imu_in + (size_t)(gk >> 6) * IMU_IN * TILE + (gk & 63);
#ifdef VF_K1_NTLOAD
#define IN(f) __builtin_nontemporal_load(in + (size_t)(f) * TILE)
#else
#define IN(f) in[(size_t)(f) * TILE]
#endif
    // the square-root information (120 of the 222 input words, read once) non-temporal: more of the bias Jacobians, which
    // are read twice, survive in L2 until their second use (PMC: 2.15 -> 2.00 GB read per launch; time unchanged)
#define INR(f) __builtin_nontemporal_load(in + (size_t)(f) * TILE)
    struct RRef { Sink& s; int a; VF_DI void operator=(double x) const { s.r(a, x); } };
    struct JRef { Sink& s; int row, col; VF_DI void operator=(double x) const { s.j(row, col, x); } };
#define OUT(f) (RRef{sink, (f)})
#define JOUT(r, c) (JRef{sink, (r), (c)})

    const State si = load_state(v, b, gk - 1), sj = load_state(v, b, gk);
    const double dt = IN(0);
    const V3 dba = si.ba - v3(IN(10), IN(11), IN(12));
    const V3 dbg = si.bg - v3(IN(13), IN(14), IN(15));
    double Hb[54];
    if constexpr (Sink::keep_h) {
#pragma unroll
        for (int i = 0; i < 54; i++) Hb[i] = IN(16 + i);
    }
#define HB(i) (Sink::keep_h ? Hb[i] : IN(16 + (i)))
    // bias-corrected preintegrated delta: d + H (b_i - bhat)   (biasCorrectedDelta)
    double xt[9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
        double s = IN(1 + r);
        s = fma(HB(r * 6 + 0), dba.x, s);
        s = fma(HB(r * 6 + 1), dba.y, s);
        s = fma(HB(r * 6 + 2), dba.z, s);
        s = fma(HB(r * 6 + 3), dbg.x, s);
        s = fma(HB(r * 6 + 4), dbg.y, s);
        s = fma(HB(r * 6 + 5), dbg.z, s);
        xt[r] = s;
    }
    const V3 tht = v3(xt[0], xt[1], xt[2]), pt = v3(xt[3], xt[4], xt[5]), vt = v3(xt[6], xt[7], xt[8]);
    const M3 Ri = qrot(si.q), Rj = qrot(sj.q);
    const V3 grav = v3(v.grav[0], v.grav[1], v.grav[2]);
    const V3 gib = mulT(Ri, grav), vib = mulT(Ri, si.vel);
    // NavState::correctPIM
    const V3 xp = pt + dt * vib + (0.5 * dt * dt) * gib;
    const V3 xv = vt + dt * gib;
    // NavState::retract -> predicted state j
    const Q4 eq = qexp(tht);
    const Q4 qp = qmul(si.q, eq);
    const V3 pp = si.t + mul(Ri, xp);
    const V3 vp = si.vel + mul(Ri, xv);
    // NavState::localCoordinates(state_j, predicted)
    const V3 rth = qlog(qmul(qconj(sj.q), qp));
    const V3 rp = mulT(Rj, pp - sj.t);
    const V3 rv = mulT(Rj, vp - sj.vel);
    const V3 rba = si.ba - sj.ba, rbg = si.bg - sj.bg;

    // closed-form 3x3 blocks of the unwhitened Jacobian (derivation: DESIGN.md "K1")
    const M3 L = so3_jr_inv(rth);
    const M3 Em = qrot(eq);
    const M3 M1 = mulBT(L, Em);            // d r_theta / d theta_i = L E^T
    const M3 Rji = mulTA(Rj, Ri);          // R_j^T R_i
    const M3 P1 = mulSkew(Rji, neg(pt));   // -R_ji [p~]x
    const M3 V1 = mulSkew(Rji, neg(vt));   // -R_ji [v~]x
    const M3 M5 = mul(L, so3_jr(tht));     // L J_r(theta~)

    // R11 = R[0:9,0:9] stays in registers; R12 / R22 columns are streamed for the bias columns
    double R[45];
#pragma unroll
    for (int a = 0; a < 9; a++)
#pragma unroll
        for (int c = a; c < 9; c++) R[idx9(a, c)] = INR(70 + off15(a) + (c - a));

    double u[9], o[9], rw[15];
    {
        const double r9[9] = {rth.x, rth.y, rth.z, rp.x, rp.y, rp.z, rv.x, rv.y, rv.z};
        double t9[9];
        white9<0, 9>(R, r9, t9);
#pragma unroll
        for (int a = 0; a < 9; a++) rw[a] = t9[a];
#pragma unroll
        for (int a = 9; a < 15; a++) rw[a] = 0.0;
    }

    // columns 0..17: X_i(theta,p) V_i X_j(theta,p) V_j ; rows 9..14 are structurally zero
#pragma unroll
    for (int c = 0; c < 3; c++) {
        // X_i.theta
#pragma unroll
        for (int r = 0; r < 3; r++) { u[r] = col3(M1, r, c); u[3 + r] = col3(P1, r, c); u[6 + r] = col3(V1, r, c); }
        white9<0, 9>(R, u, o);
#pragma unroll
        for (int a = 0; a < 9; a++) JOUT(a, c) = o[a];
        // X_i.p : rows p = R_ji
#pragma unroll
        for (int r = 0; r < 3; r++) { u[r] = 0; u[3 + r] = col3(Rji, r, c); u[6 + r] = 0; }
        white9<3, 6>(R, u, o);
#pragma unroll
        for (int a = 0; a < 6; a++) JOUT(a, 3 + c) = o[a];       // rows 6..8: structural zeros, never written
        // V_i : rows p = dt R_j^T, rows v = R_j^T
#pragma unroll
        for (int r = 0; r < 3; r++) { u[r] = 0; u[3 + r] = dt * Rj.a[c * 3 + r]; u[6 + r] = Rj.a[c * 3 + r]; }
        white9<3, 9>(R, u, o);
#pragma unroll
        for (int a = 0; a < 9; a++) JOUT(a, 6 + c) = o[a];
        // X_j.theta : rows theta = -L^T, p = [rp]x, v = [rv]x
        {
            const M3 Sp = skew(rp), Sv = skew(rv);
#pragma unroll
            for (int r = 0; r < 3; r++) { u[r] = -L.a[c * 3 + r]; u[3 + r] = col3(Sp, r, c); u[6 + r] = col3(Sv, r, c); }
        }
        white9<0, 9>(R, u, o);
#pragma unroll
        for (int a = 0; a < 9; a++) JOUT(a, 9 + c) = o[a];
        // X_j.p : rows p = -I  => -R11(:, 3+c)
#pragma unroll
        for (int a = 0; a < 9; a++)
            if (a <= 3 + c) JOUT(a, 12 + c) = -R[idx9(a < 3 + c ? a : 3 + c, 3 + c)];   // rows below: structural zeros
        // V_j : rows v = -R_j^T
#pragma unroll
        for (int r = 0; r < 3; r++) { u[r] = 0; u[3 + r] = 0; u[6 + r] = -Rj.a[c * 3 + r]; }
        white9<6, 9>(R, u, o);
#pragma unroll
        for (int a = 0; a < 9; a++) JOUT(a, 15 + c) = o[a];
    }
    // Structural zeros of the whitened Jacobian are NOT written: rows 9..14 of the columns 0..17, rows 6..8 of X_i.p,
    // the rows below the diagonal of X_j.p and rows >= 10 + c of the bias columns -- 159 of the 450 entries.  The
    // output buffers are zero-filled when the engine is created and nothing else writes those words, so readers
    // (K3, the read-backs) see zeros; a third of the J write traffic is gone.

    // bias columns: B_i (18..23) and B_j (24..29)
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const int n_rc = 10 + c;  // rows 0..9+c of R(:, 9+c) are nonzero
        double Rc[15];
#pragma unroll
        for (int a = 0; a < 15; a++) Rc[a] = (a < n_rc) ? INR(70 + off15(a) + (9 + c - a)) : 0.0;
        const V3 hth = v3(HB(0 * 6 + c), HB(1 * 6 + c), HB(2 * 6 + c));
        const V3 hp = v3(HB(3 * 6 + c), HB(4 * 6 + c), HB(5 * 6 + c));
        const V3 hv = v3(HB(6 * 6 + c), HB(7 * 6 + c), HB(8 * 6 + c));
        const V3 u0 = mul(M5, hth), u1 = mul(Rji, hp), u2 = mul(Rji, hv);
        u[0] = u0.x; u[1] = u0.y; u[2] = u0.z; u[3] = u1.x; u[4] = u1.y; u[5] = u1.z; u[6] = u2.x; u[7] = u2.y; u[8] = u2.z;
        white9<0, 9>(R, u, o);
        const double rb = (c < 3) ? vget(rba, c) : vget(rbg, c - 3);
#pragma unroll
        for (int a = 0; a < 15; a++) {
            const double bi = (a < 9) ? o[a] + Rc[a] : Rc[a];
            if (a < n_rc) {                       // rows >= 10 + c: structural zeros
                JOUT(a, 18 + c) = bi;
                JOUT(a, 24 + c) = -Rc[a];
            }
            rw[a] = fma(Rc[a], rb, rw[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 15; a++) OUT(a) = rw[a];
#undef IN
#undef INR
#undef HB
#undef OUT
#undef JOUT
}

// K1 proper.  SH (time-sharded windows): every rank evaluates the residual of EVERY factor -- the cost of a trial is then
// known on every rank without an exchange -- but writes the Jacobian only of the factors that feed the rows of H it assembles.
template <bool SH>
__device__ __forceinline__ void linearize_imu_factor(const View& v, int which, const long gk) {
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    // (the window's scalars are requested together, in front of the first branch: one memory round trip instead of one
    // per test -- with one wave per SIMD nothing else hides them)
    const int w_lo = v.lo[w], w_hi = v.hi[w], w_sel = v.sel[w], w_done = v.stop_on ? v.done[w] : 0;
    if (k <= w_lo || k >= w_hi || w_done) return;
    if (v.relin_only && !v.relin[w]) return;
    if (v.inc_on && k < v.inc_k[w]) return;      // incremental update: both keyframes of the factor are where they were
    const bool jac = !SH || !shard_skips_factor(v, w, k);
    const int b = w_sel ^ which;
    double* jbuf = v.imu_j + (size_t)b * (size_t)(v.G >> JT_LOG) * JT_STRIDE;
    HbmSink sink;
    sink.out_r = v.imu_r + ((size_t)b * (size_t)(v.G >> 6) + (size_t)(gk >> 6)) * IMU_R * TILE + (gk & 63);
    sink.slot = jbuf + (size_t)(gk >> JT_LOG) * JT_STRIDE + (gk & (JT - 1)) * 2;
    sink.jac = jac;
    linearize_imu_core(v, b, gk, sink);
}

// ------------------------------------------------------------------------------------ K2
// Algorithmic traffic per factor: 42 doubles in (2 poses x 7, record 28), 78 out.
__global__ void __launch_bounds__(VF_K1_BLOCK, VF_K1_WAVES) k_linearize_imu(View v, int which) {
    linearize_imu_factor<false>(v, which, (long)blockIdx.x * VF_K1_BLOCK + threadIdx.x);
}

// BetweenFactor<Pose3> between the keyframes in slots ga -> gk at the states of buffer b: whitened residual (6) and
// Jacobians (Ja 36, Jb 36) written to out[f * ostride], record read from in[f * istride].  `jac` = false: residual only.
template <bool SH>
__device__ __forceinline__ void between_core(const View& v, const int b, const long ga, const long gk,
                                             const double* __restrict__ in, const size_t istride,
                                             double* __restrict__ out, const size_t ostride, const bool jac) {
#define IN(f) in[(size_t)(f) * istride]
    struct NtRef { double* p; VF_DI void operator=(double x) const { __builtin_nontemporal_store(x, p); } };
    struct NtJac { double* p; bool on; VF_DI void operator=(double x) const { if (!SH || on) __builtin_nontemporal_store(x, p); } };
#define OUT(f) (NtRef{out + (size_t)(f) * ostride})
#define JOUT(f) (NtJac{out + (size_t)(f) * ostride, jac})

    const Q4 qa = q4(XS(b, 0, ga), XS(b, 1, ga), XS(b, 2, ga), XS(b, 3, ga));
    const V3 ta = v3(XS(b, 4, ga), XS(b, 5, ga), XS(b, 6, ga));
    const Q4 qb = q4(XS(b, 0, gk), XS(b, 1, gk), XS(b, 2, gk), XS(b, 3, gk));
    const V3 tb = v3(XS(b, 4, gk), XS(b, 5, gk), XS(b, 6, gk));
    const Q4 qm = q4(IN(0), IN(1), IN(2), IN(3));
    const V3 tm = v3(IN(4), IN(5), IN(6));

    // hx = T_a^-1 T_b ; err = measured^-1 hx ; r = Logmap(err)
    const M3 Ra = qrot(qa), Rm = qrot(qm);
    const Q4 qh = qmul(qconj(qa), qb);
    const V3 th = mulT(Ra, tb - ta);
    const Q4 qe = qmul(qconj(qm), qh);
    const V3 te = mulT(Rm, th - tm);
    const Xi6 xi = se3_log(qe, te);
    M3 Jw, Q2;
    se3_jr_inv(xi, &Jw, &Q2);
    // Jb = Hlocal ; Ja = -Hlocal Ad(hx^-1) = -[[JR, 0], [Q2 Rh^T - JR [th]x, JR]], JR = Jw Rh^T
    const M3 Rh = qrot(qh);
    const M3 JR = mulBT(Jw, Rh);
    const M3 QR = mulBT(Q2, Rh);
    const M3 JRS = mulSkew(JR, th);

    double Rw[21];
#pragma unroll
    for (int i = 0; i < 21; i++) Rw[i] = IN(7 + i);
    const double ru[6] = {xi.w.x, xi.w.y, xi.w.z, xi.u.x, xi.u.y, xi.u.z};
#pragma unroll
    for (int r = 0; r < 6; r++) {
        double s = 0.0;
#pragma unroll
        for (int c = r; c < 6; c++) s = fma(Rw[off6(r) + c - r], ru[c], s);
        OUT(r) = s;
    }
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double ua[6], ub[6];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            if (c < 3) {
                ua[r] = -col3(JR, r, c);
                ua[3 + r] = -(col3(QR, r, c) - col3(JRS, r, c));
                ub[r] = col3(Jw, r, c);
                ub[3 + r] = col3(Q2, r, c);
            } else {
                ua[r] = 0.0;
                ua[3 + r] = -col3(JR, r, c - 3);
                ub[r] = 0.0;
                ub[3 + r] = col3(Jw, r, c - 3);
            }
        }
#pragma unroll
        for (int r = 0; r < 6; r++) {
            double sa = 0.0, sb = 0.0;
#pragma unroll
            for (int l = r; l < 6; l++) {
                sa = fma(Rw[off6(r) + l - r], ua[l], sa);
                sb = fma(Rw[off6(r) + l - r], ub[l], sb);
            }
            JOUT(6 + r * 6 + c) = sa;
            JOUT(42 + r * 6 + c) = sb;
        }
    }
#undef IN
#undef OUT
#undef JOUT
}

template <bool SH>
__device__ __forceinline__ void linearize_between_factor(const View& v, int which, const long gk) {
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    const int lo = v.lo[w], w_hi = v.hi[w], w_sel = v.sel[w], w_done = v.stop_on ? v.done[w] : 0;
    const int a = v.btw_a[gk];                 // (all five requested together: one round trip)
    if (k <= lo || k >= w_hi || w_done) return;
    if (v.relin_only && !v.relin[w]) return;
    if (v.inc_on && k < v.inc_k[w]) return;
    const bool jac = !SH || !shard_skips_factor(v, w, k);
    if (a < lo || a >= k) return;
    const int b = w_sel ^ which;
    const double* __restrict__ in = v.btw_in + (size_t)(gk >> 6) * BTW_IN * TILE + (gk & 63);
    double* __restrict__ out = v.btw_out + ((size_t)b * (size_t)(v.G >> 6) + (size_t)(gk >> 6)) * BTW_OUT * TILE + (gk & 63);
    between_core<SH>(v, b, (long)w * v.M + a, gk, in, TILE, out, TILE, jac);
}

// ---- "far" between factors (View::x_*): BetweenFactor<Pose3> on ANY pair of keyframes of a window -- a span wider than
// the band, or a second factor on an end key (loop closures; GraphManager.cpp:83-88 takes any pair of keys).  They stay
// out of the banded H: the solve treats them as a low-rank correction (launch_extra_* below, vf_engine_solve).
// A window's slots: first its LINEAR far factors (View::xl_*: far factors whose older keyframe has been marginalised), then
// the entries of x_a / x_b.  FarRef resolves a slot: kind -1 = empty or reaching outside the window (no cost, no rows).
VF_DI void marg_delta(const View& v, int w, int b, double (&d)[27]);
#include "vf_far.hpp"
// One lane per (window, index): entry `index` of the window's nonlinear list, and far end `index` of its linear far factor
// (the six rows that belong to it)
__global__ void __launch_bounds__(64) k_linearize_extra(View v, int which) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= v.B * v.x_max) return;
    const int w = i / v.x_max, s = i - w * v.x_max;
    if (window_done(v, w)) return;
    const int b = v.sel[w] ^ which, lo = v.lo[w], hi = v.hi[w];
    {
        const int a = v.x_a[i], kb = v.x_b[i];
        double* out = v.x_out + ((size_t)b * v.B * v.x_max + i) * BTW_OUT;
        if (a < lo || kb >= hi || a >= kb) { for (int f = 0; f < BTW_OUT; f++) out[f] = 0.0; }
        else between_core<false>(v, b, (long)w * v.M + a, (long)w * v.M + kb, v.x_in + (size_t)i * BTW_IN, 1, out, 1, true);
    }
    double* lout = v.xl_out + ((size_t)b * v.B + w) * 6 * v.x_max + 6 * s;
    const int nl = v.xl_n[w];
    if (s < nl && hi - lo > 3 && v.mp_on[w]) {
        double dh[27];
        marg_delta(v, w, b, dh);
        const double* U = v.xl_U + ((size_t)w * 6 * v.x_max + 6 * s) * XL_LD;
        double acc[6];
        for (int j = 0; j < 6; j++) {
            acc[j] = v.xl_r0[(size_t)w * 6 * v.x_max + 6 * s + j];
            for (int c = 0; c < 27; c++) acc[j] = fma(U[j * XL_LD + c], dh[c], acc[j]);
        }
        for (int e = 0; e < nl; e++) {
            const double* xb = v.xl_bx + ((size_t)w * v.x_max + e) * 7;
            const State st = load_state(v, b, (long)w * v.M + v.xl_b[w * v.x_max + e]);
            const Q4 qb = q4(xb[0], xb[1], xb[2], xb[3]);
            const Xi6 xi = se3_log(qmul(qconj(qb), st.q), mulT(qrot(qb), st.t - v3(xb[4], xb[5], xb[6])));
            const double d6[6] = {xi.w.x, xi.w.y, xi.w.z, xi.u.x, xi.u.y, xi.u.z};
            for (int j = 0; j < 6; j++)
                for (int c = 0; c < 6; c++) acc[j] = fma(U[j * XL_LD + 27 + 6 * e + c], d6[c], acc[j]);
        }
        for (int j = 0; j < 6; j++) lout[j] = acc[j];
    } else
        for (int j = 0; j < 6; j++) lout[j] = 0.0;
}
// g += J^T r of the far factors, for the windows whose rows K3 has just rewritten (same test as k_assemble; engines that
// hold far factors never warm-start, so "rewritten" means the whole window)
__global__ void __launch_bounds__(64) k_extra_gradient(View v) {
    const int w = blockIdx.x, lane = threadIdx.x;
    if (!v.fresh[w] || window_done(v, w)) return;
    const int b = v.sel[w];
    for (int s = 0; s < v.x_max; s++) {              // sequential over the slots: two factors may share a keyframe
        const FarRef f = far_ref(v, w, s);
        const int nc = f.kind >= 0 ? far_cols(f) : 0;
        for (int c0 = 0; c0 < nc; c0 += 64) {        // (and over the column groups: two far ends may be the same keyframe)
            const int c = c0 + lane;
            if (c < nc) {
                int k, d;
                far_col(v, w, f, c, k, d);
                double acc = 0.0;
                for (int r = 0; r < 6; r++) acc = fma(far_jac(v, w, f, b, r, c), far_res(v, w, f, b, r), acc);
                bool first = true;                   // of the columns that land on this (keyframe, dof): far-end columns only
                if (f.kind == 1 && c >= 27)
                    for (int c2 = 27 + d; c2 < nc; c2 += 6) {
                        if (c2 == c) continue;
                        int k2, d2;
                        far_col(v, w, f, c2, k2, d2);
                        if (k2 != k) continue;
                        if (c2 < c) { first = false; break; }
                        for (int r = 0; r < 6; r++) acc = fma(far_jac(v, w, f, b, r, c2), far_res(v, w, f, b, r), acc);
                    }
                if (first) v.gvec[((size_t)w * v.M + k) * 15 + d] += acc;
            }
            __syncthreads();
        }
    }
}
// right-hand side number (s, j) of the low-rank correction: row j of the far factor in slot s of every window, scattered
// over the keyframes it touches, into a zeroed increment-shaped buffer
__global__ void __launch_bounds__(64) k_extra_rhs(View v, int s, int j, double* __restrict__ gtmp) {
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w >= v.B || window_done(v, w)) return;
    const FarRef f = far_ref(v, w, s);
    if (f.kind < 0) return;
    const int b = v.sel[w], nc = far_cols(f);
    for (int c = 0; c < nc; c++) {
        int k, d;
        far_col(v, w, f, c, k, d);
        gtmp[((size_t)w * v.M + k) * 15 + d] += far_jac(v, w, f, b, j, c);
    }
}
// ---- the Woodbury columns of a SINGLE-window engine as one batch (vf_engine.hip "far_columns"): window q of the view `c` (an
// engine of 6 x MAX_EXTRA windows with the same capacity) is a copy of the window's block rows of H with column q of U as its
// right-hand side, so that ONE partitioned solve of `c` returns every column of Z where the window's own solver would be run
// once per column.  Same kernels on the same numbers: the same bits.
__global__ void __launch_bounds__(256) k_cols_prepare(View v, View c, int ncols) {
    const int k = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
    const int lo = v.lo[0], hi = v.hi[0];
    if (window_done(v, 0)) ncols = 0;            // (the termination rule has finished the window: its column windows are empty too)
    // (every window of `c` takes part in the launch -- the separator arrays are laid out by its window count -- the ones beyond
    // the columns in use as empty windows)
    // (mp_on: the solver asks it whether row lo + 2 carries the marginal prior's coupling to lo, the block at H_DX)
    if (k == 0 && tid == 0) { c.lo[q] = q < ncols ? lo : 0; c.hi[q] = q < ncols ? hi : 0; c.lambda[q] = v.lambda[0]; c.fail[q] = 0; c.mp_on[q] = v.mp_on[0]; }
    if (k < lo || k >= hi || q >= ncols) return;
    const double* __restrict__ src = v.H + (size_t)k * HROW;
    double* __restrict__ dst = c.H + ((size_t)q * c.M + k) * HROW;
    for (int i = tid; i < HROW; i += 256) dst[i] = src[i];
    if (tid < 15) c.gvec[((size_t)q * c.M + k) * 15 + tid] = 0.0;
}
__global__ void __launch_bounds__(64) k_cols_rhs(View v, View c, int ncols) {
    const int q = threadIdx.x;
    if (q >= ncols || window_done(v, 0)) return;
    const FarRef f = far_ref(v, 0, q / 6);
    if (f.kind < 0) return;
    const int b = v.sel[0], nc = far_cols(f), j = q % 6;
    for (int col = 0; col < nc; col++) {
        int k, d;
        far_col(v, 0, f, col, k, d);
        c.gvec[((size_t)q * c.M + k) * 15 + d] += far_jac(v, 0, f, b, j, col);
    }
}
__global__ void __launch_bounds__(64) k_cols_fail(View v, View c, int ncols) {
    const int q = threadIdx.x;
    if (q < ncols && !window_done(v, 0) && c.fail[q]) v.fail[0] = 1;
}
void launch_cols_prepare(const View& v, const View& c, int ncols, hipStream_t s) {
    hipLaunchKernelGGL(k_cols_prepare, dim3((unsigned)v.M, (unsigned)c.B), dim3(256), 0, s, v, c, ncols);
    hipLaunchKernelGGL(k_cols_rhs, dim3(1), dim3(64), 0, s, v, c, ncols);
}
void launch_cols_fail(const View& v, const View& c, int ncols, hipStream_t s) {
    hipLaunchKernelGGL(k_cols_fail, dim3(1), dim3(64), 0, s, v, c, ncols);
}
// delta = y - Z (I + U^T Z)^-1 U^T y  (Woodbury; A = H_band + lambda I, y = -A^-1 g in v.delta, column q of Zm = -A^-1 u_q as
// the band solver returned it for the right-hand side u_q): one workgroup per window, the m = 6 x_max square system in LDS.
__global__ void __launch_bounds__(256) k_extra_combine(View v, const double* __restrict__ Zm, size_t zstride, int slots) {
    constexpr int MM = 6 * MAX_EXTRA;
    __shared__ double C[MM][MM + 1];
    __shared__ double cvec[MM];
    const int w = blockIdx.x, tid = threadIdx.x;
    if (v.hi[w] - v.lo[w] <= 0 || window_done(v, w)) return;
    const int m = 6 * slots, lo = v.lo[w], hi = v.hi[w], b = v.sel[w];      // slots in use (the rest are empty in every window)
    // C[p][q] = delta_pq - u_p . Zm_q  (= delta_pq + u_p^T A^-1 u_q), C[p][m] = u_p . y
    for (int e = tid; e < m * (m + 1); e += 256) {
        const int p = e / (m + 1), q = e - p * (m + 1);
        const int s = p / 6, j = p - 6 * s;
        const FarRef f = far_ref(v, w, s);
        double acc = 0.0;
        if (f.kind >= 0) {
            const double* col = (q < m ? Zm + (size_t)q * zstride : v.delta) + (size_t)w * v.M * 15;
            const int nc = far_cols(f);
            for (int c = 0; c < nc; c++) {
                int k, d;
                far_col(v, w, f, c, k, d);
                acc = fma(far_jac(v, w, f, b, j, c), col[(size_t)k * 15 + d], acc);
            }
        }
        C[p][q] = q < m ? (p == q ? 1.0 : 0.0) - acc : acc;
    }
    __syncthreads();
    // Gaussian elimination (C is symmetric positive definite: I + U^T A^-1 U), entry-parallel
    for (int c = 0; c < m; c++) {
        const double inv = 1.0 / C[c][c];
        __syncthreads();
        const int rem = m - 1 - c;
        for (int e = tid; e < rem * (rem + 1); e += 256) {
            const int p = c + 1 + e / (rem + 1), q = c + 1 + (e - (e / (rem + 1)) * (rem + 1));
            C[p][q] = fma(-C[p][c] * inv, C[c][q], C[p][q]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        for (int p = m - 1; p >= 0; p--) {
            double t = C[p][m];
            for (int q = p + 1; q < m; q++) t = fma(-C[p][q], cvec[q], t);
            cvec[p] = t / C[p][p];
        }
    }
    __syncthreads();
    // delta = y + sum_q c_q Zm_q over the window's keyframes
    for (int e = tid; e < (hi - lo) * 15; e += 256) {
        const size_t o = ((size_t)w * v.M + lo) * 15 + e;
        double acc = v.delta[o];
        for (int q = 0; q < m; q++) acc = fma(cvec[q], Zm[(size_t)q * zstride + o], acc);
        v.delta[o] = acc;
    }
}

// ------------------------------------------------------------------------------------ K2b
// delta of the marginal prior: [Local(xbar0 -> x_lo) (15); pose Local for lo+1 (6); lo+2 (6)]
VF_DI void marg_delta(const View& v, int w, int b, double (&d)[27]) {
    const int lo = v.lo[w];
    const double* xb = v.mp_x + (size_t)w * 48;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const long gk = (long)w * v.M + lo + j;
        const State s = load_state(v, b, gk);
        const Q4 qb = q4(xb[16 * j], xb[16 * j + 1], xb[16 * j + 2], xb[16 * j + 3]);
        const V3 tb = v3(xb[16 * j + 4], xb[16 * j + 5], xb[16 * j + 6]);
        const Xi6 xi = se3_log(qmul(qconj(qb), s.q), mulT(qrot(qb), s.t - tb));
        const int o = j == 0 ? 0 : 15 + 6 * (j - 1);
        d[o] = xi.w.x; d[o + 1] = xi.w.y; d[o + 2] = xi.w.z; d[o + 3] = xi.u.x; d[o + 4] = xi.u.y; d[o + 5] = xi.u.z;
        if (j == 0) {
            d[6] = s.vel.x - xb[7]; d[7] = s.vel.y - xb[8]; d[8] = s.vel.z - xb[9];
            d[9] = s.ba.x - xb[10]; d[10] = s.ba.y - xb[11]; d[11] = s.ba.z - xb[12];
            d[12] = s.bg.x - xb[13]; d[13] = s.bg.y - xb[14]; d[14] = s.bg.z - xb[15];
        }
    }
}

__global__ void __launch_bounds__(256) k_linearize_between(View v, int which) {
    linearize_between_factor<false>(v, which, (long)blockIdx.x * 256 + threadIdx.x);
}

__device__ __forceinline__ void linearize_prior_window(const View& v, int which, const int w) {
    if (w >= v.B || window_done(v, w)) return;
    if (v.relin_only && !v.relin[w]) return;
    if (v.inc_on && !v.inc_prior && v.inc_k[w] > v.lo[w] + 2) return;   // (the priors sit on the window's first three keyframes)
    const int b = v.sel[w] ^ which;
    if (v.mp_on[w] && v.hi[w] - v.lo[w] >= 3) {
        // marginal prior: gm = L d + eta, cost = 0.5 d^T L d + eta^T d (fixed linearisation point)
        double d[27];
        marg_delta(v, w, b, d);
        const double* L = v.mp_L + (size_t)w * 729;
        const double* eta = v.mp_eta + (size_t)w * 27;
        double* out = v.mp_out + ((size_t)b * v.B + w) * 28;
        double cost = 0.0;
        for (int i = 0; i < 27; i++) {
            double g = 0.0;
            for (int j = 0; j < 27; j++) g = fma(L[i * 27 + j], d[j], g);
            cost += d[i] * (0.5 * g + eta[i]);
            out[i] = g + eta[i];
        }
        out[27] = cost;
    }
    const int k = v.prior_k[w];
    if (k < v.lo[w] || k >= v.hi[w]) return;
    const long gk = (long)w * v.M + k;
    const double* in = v.prior_in + (size_t)w * PRIOR_IN;
    double* out = v.prior_out + ((size_t)b * v.B + w) * PRIOR_OUT;
    const State s = load_state(v, b, gk);
    const Q4 qp = q4(in[0], in[1], in[2], in[3]);
    const V3 tp = v3(in[4], in[5], in[6]);
    const M3 Rp = qrot(qp);
    const Xi6 xi = se3_log(qmul(qconj(qp), s.q), mulT(Rp, s.t - tp));
    M3 Jw, Q2;
    se3_jr_inv(xi, &Jw, &Q2);
    const double* sig = in + 16;
    const double r6[6] = {xi.w.x, xi.w.y, xi.w.z, xi.u.x, xi.u.y, xi.u.z};
    for (int i = 0; i < 225; i++) out[15 + i] = 0.0;
    for (int r = 0; r < 3; r++) {
        out[r] = r6[r] / sig[r];
        out[3 + r] = r6[3 + r] / sig[3 + r];
        for (int c = 0; c < 3; c++) {
            out[15 + r * 15 + c] = Jw.a[r * 3 + c] / sig[r];
            out[15 + (3 + r) * 15 + c] = Q2.a[r * 3 + c] / sig[3 + r];
            out[15 + (3 + r) * 15 + 3 + c] = Jw.a[r * 3 + c] / sig[3 + r];
        }
    }
    const double xs[9] = {s.vel.x, s.vel.y, s.vel.z, s.ba.x, s.ba.y, s.ba.z, s.bg.x, s.bg.y, s.bg.z};
    for (int i = 0; i < 9; i++) {
        out[6 + i] = (xs[i] - in[7 + i]) / sig[6 + i];
        out[15 + (6 + i) * 15 + 6 + i] = 1.0 / sig[6 + i];
    }
}
__global__ void k_linearize_prior(View v, int which) {
    linearize_prior_window(v, which, blockIdx.x * blockDim.x + threadIdx.x);
}
// K2 + K2b in one launch (large batches): the prior linearisations are one lane per window, i.e. a handful of waves
// running an 80 us latency chain; as the first workgroups of K2's grid they hide behind the between factors
__global__ void __launch_bounds__(256) k_linearize_between_prior(View v, int which, int nb_pri) {
    const int bx = blockIdx.x;
    if (bx < nb_pri) {      // one wave of windows per workgroup: four such chains on one CU ran 1.6 x longer
        if (threadIdx.x < 64) linearize_prior_window(v, which, bx * 64 + (int)threadIdx.x);
    } else linearize_between_factor<false>(v, which, (long)(bx - nb_pri) * 256 + threadIdx.x);
}
// K1 + K2 + K2b in ONE launch, for few windows (latency form): with a handful of windows each of the three kernels
// is a single latency chain (27 / 9 / 13 us), so running them side by side saves two of the three; for large
// batches they stay separate (K2 would inherit K1's register footprint here).
// (SH: the form time-sharded windows use, whatever the batch size)
template <bool SH>
__global__ void __launch_bounds__(VF_K1_BLOCK, VF_K1_WAVES) k_linearize_all(View v, int which, int nb_imu, int nb_btw) {
    const int bx = blockIdx.x;
    if (bx < nb_imu) linearize_imu_factor<SH>(v, which, (long)bx * VF_K1_BLOCK + threadIdx.x);
    else if (bx < nb_imu + nb_btw) linearize_between_factor<SH>(v, which, (long)(bx - nb_imu) * VF_K1_BLOCK + threadIdx.x);
    else linearize_prior_window(v, which, (bx - nb_imu - nb_btw) * VF_K1_BLOCK + (int)threadIdx.x);
}

// Warm start of a fixed-lag update (vf_engine_slide directly after a solve): the linearisation of every factor that was
// in the window is still the one of the current states (an accepted trial's records became current with its states, a
// rejected one left both alone), so only the factors of the `nslid` appended keyframes and the priors (the marginal prior
// has just changed) are linearised.  H and g are then stale only at the two ends of a window whose last trial was
// rejected: fresh[w] = 1 + nslid tells k_assemble to redo just those tiles (fresh[w] = 1: all of them).
__global__ void __launch_bounds__(VF_K1_BLOCK, VF_K1_WAVES) k_linearize_tail(View v, int nslid) {
    const int bx = blockIdx.x, t = threadIdx.x;
    if (bx < v.B) {
        if (t < nslid) linearize_imu_factor<false>(v, 0, (long)bx * v.M + v.hi[bx] - 1 - t);
    } else if (bx < 2 * v.B) {
        const int w = bx - v.B;
        if (t < nslid) linearize_between_factor<false>(v, 0, (long)w * v.M + v.hi[w] - 1 - t);
    } else {
        const int w = (bx - 2 * v.B) * 64 + t;      // one wave of windows per workgroup
        if (t < 64) {
            if (w < v.B) v.fresh[w] = v.fresh[w] ? 1 : 1 + nslid;
            linearize_prior_window(v, 0, w);
        }
    }
}

// ------------------------------------------------------------------------------------ K3
// Block-sparse J^T J / J^T r, owner-computes per keyframe, deterministic (no atomics):
//   H[k][k]   = Jj^T Jj (imu k) + Ji^T Ji (imu k+1) + between (as b at k, as a at k+d) + prior
//   H[k][k-1] = Jj^T Ji (imu k) + Jb^T Ja (between a=k-1)
//   H[k][k-d] = Jb^T Ja (between a=k-d), pose 6x6 only
// One 256-thread block per tile of 16 keyframes.  The tile's 17 IMU linearisations (r | J) are
// staged once into LDS with coalesced 128-byte segments (every J entry leaves HBM exactly once),
// then each wave forms the 16x16 blocks of [J r]^T [J r] for 4 keyframes with
// v_mfma_f64_16x16x4 (column 15 of a tile carries J^T r, so the gradient comes for free) and
// writes the H block rows coalesced straight from the MFMA C layout.  The 6x6 between terms and
// the prior are added on the VALU from LDS before the store.
// structural zeros of the whitened 15x30 IMU Jacobian (k_linearize_imu never writes them, the buffers are zero-filled
// at creation): field f = 15 + 30 r + c of a factor's (r | J) record
__host__ __device__ constexpr bool imu_field_is_zero(int f) {
    if (f < 15 || f >= IMU_OUT) return false;
    const int r = (f - 15) / 30, c = (f - 15) % 30;
    if (c < 18) return r >= 9 || (c >= 3 && c < 6 && r >= 6) || (c >= 12 && c < 15 && r > 3 + (c - 12));
    return r >= 10 + (c - 18) % 6;
}
#ifndef VF_K3_AT
#define VF_K3_AT (1 << VF_JT_LOG)
#endif
#ifndef VF_K3_NT
#define VF_K3_NT 256
#endif
// bit b of word w: field 32 w + b of a factor's (r | J) record exists and is not a structural zero
__host__ __device__ constexpr unsigned imu_nz_word(int w) {
    unsigned m = 0;
    for (int bit = 0; bit < 32; bit++) {
        const int f = 32 * w + bit;
        if (f < IMU_OUT && !imu_field_is_zero(f)) m |= 1u << bit;
    }
    return m;
}
constexpr int AT = VF_K3_AT;    // keyframes per block
constexpr int K3_NT = VF_K3_NT; // threads per block
constexpr int K3_KPW = AT / (K3_NT / 64);   // keyframes per wave
static_assert(K3_KPW * (K3_NT / 64) == AT && K3_NT % AT == 0, "K3 tiling");
constexpr int LBS = 79;         // LDS stride of one between linearisation (78 + pad), odd

#ifdef VF_SOLVE_STAMPS   // diagnostic build only: phase time stamps of one workgroup of K3 (tools/k3_stamps_probe.py)
__device__ unsigned long long g_k3_stamps[8];
#define K3STAMP(i) do { if (blockIdx.x == 40 && blockIdx.y == (gridDim.y >> 1) && threadIdx.x == 0) { unsigned long long _t; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); g_k3_stamps[i] = _t; } } while (0)
#else
#define K3STAMP(i) do {} while (0)
#endif
#ifdef VF_SOLVE_STAMPS
__device__ unsigned long long g_k3_loop[8];
#define K3LOOP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long _t; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); __builtin_amdgcn_sched_barrier(0); k3acc[i] += _t - k3prev; k3prev = _t; } while (0)
#else
#define K3LOOP(i) do {} while (0)
#endif
// Per-lane operand maps of assemble_tile and the window scalars it needs.  They come from memory (the index table of the J
// stream, per-window flags): the caller builds them while its staging loads are in flight, not behind the barrier that
// follows them -- two dependent round trips (about 3 k cycles per tile) taken off the matrix-core phase.
struct AsmMaps { int offI[4], offJ[4], oA[2], oB[2], prior_key; bool marg_on; };
__device__ __forceinline__ AsmMaps make_asm_maps(const View& v, const int w, const int lo, const int hi, const int lane) {
    AsmMaps m;
    const int ci = lane & 15, kq = lane >> 4;
    int (&offI)[4] = m.offI, (&offJ)[4] = m.offJ, (&oA)[2] = m.oA, (&oB)[2] = m.oB;
    // MFMA operand words of this lane in a factor's LDS image: J[4 q + kq][column ci of the i / j side], column 15 = r;
    // structural zeros and the padding row 15 read the zero cell
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int row = 4 * q + kq;
        offI[q] = offJ[q] = LJ_ZERO;
        if (row < 15) {
            if (ci < 15) {
                const int eI = JMD.idx[row * 30 + imu_col(0, ci)], eJ = JMD.idx[row * 30 + imu_col(1, ci)];
                if (eI >= 0) offI[q] = eI;
                if (eJ >= 0) offJ[q] = 2 * JS_PI + eJ;
            } else offI[q] = offJ[q] = LJ_R + row;
        }
    }
    // Between-factor terms on the matrix cores as well: a between linearisation in LDS (r: 0, Ja: 6, Jb: 42, 6 rows) is
    // the 6 x 16 operand X = [J | 0 ... 0 | r] (column 15 = r, like the IMU tiles), rows padded to 8 = two k-steps;
    // X^T X adds J^T J to the pose block and J^T r to the gradient column of a diagonal tile, Xb^T Xa is the coupling
    // block.  Per lane: the in-slot offsets of its two operand words (the slot's pad cell = 0 where X has no entry).
    // (the VALU form, 6-term dot products per entry from LDS, cost 0.5 ms of K3's 2.7; this one about 0.35)
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int row = 4 * q + kq;
        const bool valid = row < 6;
        oA[q] = !valid ? BTW_OUT : (ci < 6 ? 6 + row * 6 + ci : (ci == 15 ? row : BTW_OUT));
        oB[q] = !valid ? BTW_OUT : (ci < 6 ? 42 + row * 6 + ci : (ci == 15 ? row : BTW_OUT));
    }
    m.prior_key = v.prior_k[w];
    m.marg_on = v.mp_on[w] != 0 && hi - lo >= 3;
    return m;
}
// The matrix-core part of K3: wave `wv` of the workgroup forms the block rows of KPW keyframes from the tile's (r | J) rows in LJ and the
// between linearisations in LB (s_a = their source keyframes), and stores them into H, g.
template <int KPW>
__device__ __forceinline__ void assemble_tile(const View& v, const double* __restrict__ LJ, const double* __restrict__ LB,
                                              const int* __restrict__ s_a, const int w, const int b, const int k0,
                                              const long gk0, const int lo, const int hi, const int rlo, const int rhi,
                                              const int wv, const int lane, const AsmMaps& maps) {
    const int ci = lane & 15, kq = lane >> 4;
    const int (&offI)[4] = maps.offI, (&offJ)[4] = maps.offJ, (&oA)[2] = maps.oA, (&oB)[2] = maps.oB;
    auto load_ops = [&](int lf, double (&ai)[4], double (&aj)[4]) {
        const double* F = LJ + lf * LJS;
#pragma unroll
        for (int q = 0; q < 4; q++) { ai[q] = F[offI[q]]; aj[q] = F[offJ[q]]; }
    };
#ifdef VF_SOLVE_STAMPS
    unsigned long long k3acc[8] = {0}, k3prev = __builtin_amdgcn_s_memtime();
#endif
    // (window-level scalars: read by the caller in front of its staging barrier -- a global load inside the loop would have
    // to wait for its result with s_waitcnt vmcnt(0), and stores count in vmcnt on this ISA)
    const int prior_key = maps.prior_key;
    const bool marg_on = maps.marg_on;
    d4_t D = {0, 0, 0, 0};
    const int lf0 = KPW * wv;
#pragma unroll 1
    for (int lf = lf0; lf <= lf0 + KPW; lf++) {
        K3LOOP(0);
        double ai[4], aj[4];
        load_ops(lf, ai, aj);
        K3LOOP(1);    // operands of factor lf in registers
        if (lf > lf0) {
            // ---- finish keyframe kf = lf-1: D += Ji^T Ji of factor lf, add 6x6 terms, store
#pragma unroll
            for (int q = 0; q < 4; q++) D = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[q], ai[q], D, 0, 0, 0);
            K3LOOP(2);    // Ji^T Ji issued
            const int kl = lf - 1, k = k0 + kl;
            if (k >= rlo && k < rhi) {
                const long gk = gk0 + kl;
                double* Hk = v.H + (size_t)gk * HROW;
                // (accumulated into D itself: separate accumulators added at the end measured 4 % slower)
                if (__builtin_amdgcn_readfirstlane(s_a[kl]) >= 0) {   // a between factor ends here: Jb^T [Jb | r]
                    const double x0 = LB[kl * LBS + oB[0]], x1 = LB[kl * LBS + oB[1]];
                    D = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, D, 0, 0, 0);
                    D = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, D, 0, 0, 0);
                }
#pragma unroll
                for (int d = 1; d <= 3; d++)
                    if (__builtin_amdgcn_readfirstlane(s_a[kl + d]) == k) {   // ... or starts here: Ja^T [Ja | r]
                        const double x0 = LB[(kl + d) * LBS + oA[0]], x1 = LB[(kl + d) * LBS + oA[1]];
                        D = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, D, 0, 0, 0);
                        D = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, D, 0, 0, 0);
                    }
                const bool is_prior = prior_key == k;
                const double* Pq = v.prior_out + ((size_t)b * v.B + w) * PRIOR_OUT;
                const int mo = marg_on ? k - lo : 99;   // 0,1,2: rows of the marginal prior
                const double* ML = v.mp_L + (size_t)w * 729;
                const double* Mg = v.mp_out + ((size_t)b * v.B + w) * 28;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int a = kq + 4 * r;
                    double val = D[r];
                    if (is_prior && a < 15) {
                        double sum = 0.0;
                        for (int rr = 0; rr < 15; rr++)
                            sum = fma(Pq[15 + rr * 15 + a], ci < 15 ? Pq[15 + rr * 15 + ci] : Pq[rr], sum);
                        val += sum;
                    }
                    if (mo == 0 && a < 15) val += ci < 15 ? ML[a * 27 + ci] : Mg[a];
                    if ((mo == 1 || mo == 2) && a < 6 && (ci < 6 || ci == 15)) {
                        const int ob = mo == 1 ? 15 : 21;
                        val += ci < 6 ? ML[(ob + a) * 27 + ob + ci] : Mg[ob + a];
                    }
                    if (a < 15) {
                        if (ci <= a) __builtin_nontemporal_store(val, Hk + H_D0 + h_tri(a, ci));   // lower triangle only
                        else if (ci == 15) v.gvec[(size_t)gk * 15 + a] = val;
                    }
                }
            }
        }
        K3LOOP(3);        // diagonal tile finished and stored
        if (lf < lf0 + KPW) {
            // ---- keyframe kf = lf: off-diagonal block Jj^T Ji, pose-only blocks, start D = Jj^T Jj
            d4_t O = {0, 0, 0, 0};
            D = (d4_t){0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                O = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[q], ai[q], O, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f64_16x16x4f64(aj[q], aj[q], D, 0, 0, 0);
            }
            const int kl = lf, k = k0 + kl;
            if (k >= rlo && k < rhi) {
                double* Hk = v.H + (size_t)(gk0 + kl) * HROW;
                const int ak = __builtin_amdgcn_readfirstlane(s_a[kl]);
                const int dk = ak >= 0 ? k - ak : 0;   // 1..3 when a between factor ends here
                const int mo2 = marg_on ? k - lo : 99;
                const double* ML2 = v.mp_L + (size_t)w * 729;
                d4_t T = {0, 0, 0, 0};                    // Jb^T Ja of the between factor ending here
                if (dk >= 1) {
                    const double b0 = LB[kl * LBS + oB[0]], b1 = LB[kl * LBS + oB[1]];
                    const double a0 = LB[kl * LBS + oA[0]], a1 = LB[kl * LBS + oA[1]];
                    T = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, T, 0, 0, 0);
                    T = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, T, 0, 0, 0);
                }
                if (k > lo) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int a = kq + 4 * r;
                        double val = O[r];
                        if (dk == 1 && a < 6 && ci < 6) val += T[r];
                        if (mo2 == 1 && a < 6 && ci < 15) val += ML2[(15 + a) * 27 + ci];            // (lo+1 pose) x (lo: 15)
                        if (mo2 == 2 && a < 6 && ci < 6) val += ML2[(21 + a) * 27 + 15 + ci];        // (lo+2 pose) x (lo+1 pose)
                        if (a < 15 && ci < 15) __builtin_nontemporal_store(val, Hk + H_D1 + a * 15 + ci);
                    }
                }
                // pose x pose blocks two and three keyframes back, straight from the MFMA C layout (rows kq + 4 r)
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int a6 = kq + 4 * r;
                    if (a6 < 6 && ci < 6) {
                        const double x = T[r];
                        Hk[H_D2 + a6 * 6 + ci] = (dk == 2 ? x : 0.0) + (mo2 == 2 ? ML2[(21 + a6) * 27 + ci] : 0.0);
                        Hk[H_D3 + a6 * 6 + ci] = dk == 3 ? x : 0.0;
                    }
                }
                // the marginal prior couples (lo+2 pose) with all 15 dof of lo: columns 6..14 of the
                // d=2 strip (the solver reads them only for the window's third keyframe)
                if (mo2 == 2 && lane < 54) {
                    const int a6 = lane / 9, b9 = 6 + lane - a6 * 9;
                    Hk[H_DX + a6 * 9 + (b9 - 6)] = ML2[(21 + a6) * 27 + b9];
                }
            }
        }
    }
    K3LOOP(4);        // off-diagonal part
#ifdef VF_SOLVE_STAMPS
    if (blockIdx.x == 40 && blockIdx.y == (gridDim.y >> 1) && threadIdx.x == 0) for (int i = 0; i < 8; i++) g_k3_loop[i] = k3acc[i];
#endif
}

// Four waves per SIMD (128 VGPRs) = four workgroups per CU, which is also what the 40 KB of LDS allow: 2.23 -> 1.98 ms once
// the staging code had come down to 132 VGPRs (at 176 it spilled and lost).
#ifndef VF_K3_WPE
#define VF_K3_WPE 4
#endif
__attribute__((amdgpu_waves_per_eu(VF_K3_WPE, VF_K3_WPE)))
__global__ void __launch_bounds__(K3_NT) k_assemble(View v) {
    __shared__ __attribute__((aligned(16))) double LJ[(AT + 1) * LJS];
    __shared__ double LB[(AT + 3) * LBS];
    __shared__ int s_a[AT + 3];
    const int tid = threadIdx.x;
    // grid = (tiles per window, windows).  A rejected LM trial leaves the current linearisation, hence H and g,
    // unchanged (k_decide clears `fresh` on reject; accept / init / slide set it): such a tile must cost as little
    // as a launch can -- one flag read, no index arithmetic in front of it (an all-rejected batch used to take 0.73 ms)
    const int w = blockIdx.y + v.w_first;
    // (the window's scalars requested together, in front of the first test: one memory round trip, not three)
    // (hybrid solve of an engine whose sweep assembles its own rows: K3 works for the partitioned form only -- launched with
    // gate = 2, it returns while the sweep is the form in charge, and when it does run it cannot trust `fresh`: the trials the
    // sweep served never brought H up to date)
    const int fr = (v.gate == 2 || v.inc_on) ? 1 : v.fresh[w], lo = v.lo[w], hi = v.hi[w], w_sel = v.sel[w], w_done = v.stop_on ? v.done[w] : 0;
    const int ik = v.inc_on ? v.inc_k[w] : 0;
    if (!fr || w_done || gated_off(v)) return;
    const int k0 = blockIdx.x * AT;
    const long gk0 = (long)w * v.M + k0;
    // incremental update: a factor reaches three keyframes back, so the rows from inc_k - 3 on are new -- all of them when
    // the sweep starts at the window's first keyframe (the head rows carry the marginal prior of the last slide)
    const int inc_rlo = (v.inc_on && inc_start(ik, lo) > lo) ? ik - 3 : lo;
    if (k0 + AT <= inc_rlo) return;
    // warm start (k_linearize_tail): fr = 1 + appended keyframes; only rows near the ends of the window changed --
    // head: the marginal prior / the factors that left with the oldest keyframe reach rows lo .. lo+3;
    // tail: a new factor at slot b touches rows b-3 .. b
    if (fr >= 2 && fr < 64 && k0 >= lo + 4 && k0 + AT <= hi - (fr - 1) - 4) return;
    int rlo, rhi;                            // rows of H this rank assembles (absolute slots)
    own_range(v, w, rlo, rhi);
    rlo += lo;
    rhi += lo;
    if (v.sh_G > 1 && rhi + 2 <= hi) rhi += 2;   // tail rows of the rank's last chunk (see shard_skips_factor)
    else if (v.sh_G > 1) rhi = hi;
    rlo = rlo > inc_rlo ? rlo : inc_rlo;
    if (k0 + AT <= rlo || k0 >= rhi) return; // no owned active keyframe in this tile (uniform)
    K3STAMP(0);
    const int b = w_sel;
    const size_t tiles = (size_t)(v.G >> 6);
    const double* btw_out = v.btw_out + (size_t)b * tiles * BTW_OUT * TILE;

    AsmMaps maps;
    // ---- stage the tile's 9 IMU linearisations from the J stream: the tile's own block is copied into LDS as it is
    // (146 pairs x 8 slots, 16 bytes per lane and load, every 128-B line used whole, one ds_write_b128 each), the i-side
    // pairs of the halo factor come from slot 0 of the next tile.  All of a thread's global loads are issued before the
    // first LDS write.
    {
        constexpr int NOWN = JS_PAIRS * JT, NALL = NOWN + JS_PI, NJ = (NALL + K3_NT - 1) / K3_NT,
                      NB = ((AT + 3) * BTW_OUT + K3_NT - 1) / K3_NT;
        static_assert(AT == JT, "K3 tiles are the tiles of the J stream");
        const d2_t* jt = (const d2_t*)(v.imu_j + ((size_t)b * (size_t)(v.G >> JT_LOG) + (size_t)(gk0 >> JT_LOG)) * JT_STRIDE);
        const bool halo_ok = k0 + AT > lo && k0 + AT < hi;          // (then the next tile exists: k0 + AT < M)
        d2_t tj[NJ];
#pragma unroll
        for (int it = 0; it < NJ; it++) {
            const int e = it * K3_NT + tid;
            const int src = e < NOWN ? e : (e < NALL && halo_ok ? NOWN + (e - NOWN) * JT : 0);   // halo: pair p of slot 0 of the next tile
            tj[it] = jt[src];
        }
        // residuals of the 9 factors: 135 words
        double tr = 0.0;
        if (tid < (AT + 1) * IMU_R) {
            const int fac = tid / IMU_R, a = tid - fac * IMU_R;
            const long gf = gk0 + fac;
            const int kf = k0 + fac;
            if (kf > lo && kf < hi) tr = v.imu_r[((size_t)b * tiles + (size_t)(gf >> 6)) * IMU_R * TILE + (size_t)a * TILE + (gf & 63)];
        }
        // between linearisations of slots k0 .. k0+10
        double tb[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int e = tid + K3_NT * j;
            const int sl = e / BTW_OUT, f = e - sl * BTW_OUT;
            const int ks = k0 + sl;
            const long gs = gk0 + sl;
            tb[j] = (e < (AT + 3) * BTW_OUT && ks > lo && ks < hi) ? btw_out[((size_t)(gs >> 6) * BTW_OUT + f) * TILE + (gs & 63)] : 0.0;
        }
        K3STAMP(1);   // all loads issued
        maps = make_asm_maps(v, w, lo, hi, tid & 63);      // (their table / flag reads fly with the staging loads)
        if (tid < AT + 3) {
            LB[tid * LBS + BTW_OUT] = 0.0;       // the pad cell of a slot: the zero the MFMA operand maps point at
            const int ks = k0 + tid;
            int a = -1;
            if (ks > lo && ks < hi) { a = v.btw_a[gk0 + tid]; if (a < lo || a >= ks) a = -1; }
            s_a[tid] = a;
        }
        if (tid <= AT) LJ[tid * LJS + LJ_ZERO] = 0.0;
        if (tid < (AT + 1) * IMU_R) LJ[(tid / IMU_R) * LJS + LJ_R + tid % IMU_R] = tr;
#pragma unroll
        for (int it = 0; it < NJ; it++) {
            const int e = it * K3_NT + tid;
            if (e < NALL) {
                const int own = e < NOWN;
                const int pr = own ? e >> JT_LOG : e - NOWN;
                const int fac = own ? e & (JT - 1) : AT;
                const int kf = k0 + fac;
                d2_t x = tj[it];
                if (!(kf > lo && kf < hi)) { x.x = 0.0; x.y = 0.0; }        // factor outside the window: zeros
                *(d2_t*)(LJ + fac * LJS + 2 * pr) = x;
            }
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int e = tid + K3_NT * j;
            if (e < (AT + 3) * BTW_OUT) { const int sl = e / BTW_OUT; LB[sl * LBS + (e - sl * BTW_OUT)] = tb[j]; }
        }
    }
    K3STAMP(2);   // own loads landed, LDS written
    __syncthreads();
    K3STAMP(3);   // everybody's

    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave-uniform: the keyframe loop and its tests stay scalar
    assemble_tile<K3_KPW>(v, LJ, LB, s_a, w, b, k0, gk0, lo, hi, rlo, rhi, wv, lane, maps);
    K3STAMP(4);       // wave 0: MFMAs done, stores issued
#ifdef VF_SOLVE_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    K3STAMP(5);       // ... and acknowledged
#endif
}


// ------------------------------------------------------------------------------------ K4
VF_DI double readlane_d(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// lanes 0..15 of x copied into the other three 16-lane rows: two VALU lane swaps per 32-bit half (gfx950's
// v_permlane16_swap: rows 1, 3 of the first operand <-> rows 0, 2 of the second; v_permlane32_swap: upper half <-> lower half)
VF_DI double rep_row0(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const auto c = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(b[0], b[0], false, false);
    return __hiloint2double((int)d[0], (int)c[0]);
}

// One wavefront per window.  Right-looking block Cholesky of the block-banded normal matrix,
// exploiting its profile: IMU factors couple consecutive keyframes in all 15 dof, between
// factors couple keyframes up to 3 apart in the 6 pose dof only, so the active set while
// eliminating keyframe k is  [k: 15] [k+1: 15] [k+2: pose 6] [k+3: pose 6]  = 42 rows.
// The panel of step k has 58 rows, one per lane:
//     0..14  pivot block          15..41  sub-diagonal rows        42  right-hand side
//    43..57  identity  -> after the column operations these rows hold L_kk^-T
// (the rhs row makes the forward substitution free, the identity rows turn the back substitution
// into a mat-vec instead of a 15-step triangular chain).
//   * panel factorisation in registers, as generated straight-line code in a fixed issue order (tools/gen_pivot.py,
//     vf_pivot_15.inc): the chain pivot c -> c+1 (scale, v_readlane the multiplier, update, v_readlane the next pivot,
//     v_rsq_f64 + one third-order correction) interleaved with the other columns' updates -- the two nearest by
//     v_readlane broadcasts issued one ahead of their use, the rest by one v_fmac_f64_dpp row_newbcast each; a
//     non-positive pivot shows as NaN / inf in the last reciprocal, tested once per step;
//   * Schur update of the trailing 28x27 block (27 active rows + rhs) = C - P P^T on the matrix
//     cores: 3 lower tiles x 4 k-steps of v_mfma_f64_16x16x4, operands straight from the LDS panel, the trailing
//     entries as accumulator input (negated A operands), results written back without a read-out pass;
//   * trailing window: circular 4-keyframe LDS buffer.  The k loop is unrolled by 4 so that the
//     slot arithmetic ((k+d)&3) is a compile-time constant: every LDS address is a per-lane
//     constant plus an immediate; block rows of H are fetched from HBM three steps ahead, unconditionally
//     (absent blocks come from a row of zeros);
//   * panel rows 15..57 go to HBM straight from registers, by column pairs (see "Panel" below);
//   * back substitution software-pipelined: the increment of the previous step stays in registers (v_readlane
//     broadcasts), everything that does not depend on it is prepared one step ahead.
// One-wave workgroup: LDS operations of a wave retire in issue order, so cross-lane hand-offs
// through LDS need no s_barrier and no vmcnt(0) (which __syncthreads() carries and which would
// stall every step on the in-flight HBM prefetch); a compiler barrier (+ lgkmcnt(0) where a value is used) is enough.
// Sequential in k: latency-bound for one window, HBM-bound (5 TB/s) under a full batch; see DESIGN.md "K4".
constexpr int LDW = 61;
// Panel of one keyframe in HBM (vf_kernels.hpp "Cholesky panel"): column pairs, lane = row, so one 16-byte store / load
// instruction of the sweeps covers up to 43 x 16 contiguous bytes -- with a row per 128-B line every instruction touched 43
// different lines, 16 bytes of each.  The offset of a lane's entry of pair c is a per-lane constant (its place in the pair,
// or the keyframe's zero cell where the entry is a structural zero of L^-T): no predicate in either sweep.
// a wave-uniform pointer into global memory, told to the compiler as such (SGPR base + per-lane 32-bit offset addressing)
#define VF_GLOBAL __attribute__((address_space(1)))
template <class T> VF_DI VF_GLOBAL char* uniform_gptr(T* p) {
    const unsigned long long x = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
    return (VF_GLOBAL char*)(((unsigned long long)hi << 32) | lo);
}
VF_DI int panel_pair_off(int row, int c) { return (row < panel_rows(c) ? panel_off(c) + 2 * row : PANEL_DUMP) * (int)sizeof(double); }
constexpr int S_WD = 0;                  // LDS map (doubles)
constexpr int S_GD = 60 * LDW;           // 3660: rhs, circular
constexpr int S_DUMP = S_GD + 64;        // write sink for masked-off lanes (never read)
constexpr int S_ZERO = S_DUMP + 96;      // 16 zeros: read source for masked-off lanes
constexpr int S_ID = S_ZERO + 16;        // 15x15 identity: panel rows of the L^-T lanes
constexpr int S_P = S_ID + 225;          // panel rows 15..42 at stride 15 (conflict-free column reads)
constexpr int S_DL = S_P + 43 * 15;      // back-substitution: delta of the 3 following keyframes
constexpr int S_BC = S_DL + 64;          // 16: pivot-block entries of the column just scaled, read back replicated per 16-lane row
constexpr int S_TOTAL = S_BC + 16;
// Compact trailing window of the SPLIT forward sweep (SOLVE_FULL_FWD, k_band_forward): 20.2 KB of LDS per wave instead of 38.3,
// i.e. EIGHT one-wave workgroups per CU = two resident waves per SIMD (the kernel's 252 registers allow that too).
// The 60 x 61 window above keeps every (row slot, column slot) pair of its four circular keyframe slots, although only
// the lower block triangle within the profile [k: 15][k+1: 15][k+2: pose][k+3: pose] is ever live.  Here a block is filed
// under the slot of its ROW keyframe and its DISTANCE e to the column keyframe -- both stay the same while the sweep
// moves on (row and column keyframe age together), so nothing is ever copied or promoted:
//     e = 0: lower triangle, 120      e = 1: 15 x 15      e = 2: 6 x 15 (pose rows; fill-in and the marginal prior's strip)
//     e = 3: 6 x 6 (pose x pose; a panel row of that block is read 15 wide and its columns 6..14 are masked to zero)
// the identity rows of the panel come from a 29-cell strip (0 x 14, 1, 0 x 14) read at a per-lane offset.
constexpr int CW_D0 = 0, CW_D1 = 120, CW_D2 = CW_D1 + 225, CW_D3 = CW_D2 + 90, CW_SLOT = CW_D3 + 36;
constexpr int CW_GD = 4 * CW_SLOT;           // rhs, circular (4 x 15, padded to 64)
constexpr int CW_DUMP = CW_GD + 64;
constexpr int CW_ZERO = CW_DUMP + 96;
constexpr int CW_ID = CW_ZERO + 16;          // identity strip: cell 14 = 1
constexpr int CW_P = CW_ID + 30;             // panel rows 15..42 at stride 15 (MFMA operands)
constexpr int CW_BC = CW_P + 28 * 15;
constexpr int CW_TOTAL = CW_BC + 16;
static_assert(CW_TOTAL * 8 <= 20480, "eight compact forward sweeps per CU (160 KB of LDS)");
// Assembling forward sweep (SOLVE_ASM_FWD, k_band_forward_asm): the compact window, then the LDS image of ONE tile of the J
// stream (8 factors at K3's stride LJS: pairs, residual, zero cell) and ONE between linearisation (78 words + the zero cell).
constexpr int AS_LJ = CW_TOTAL, AS_LB = AS_LJ + JT * LJS, AS_TOTAL = AS_LB + 80;
constexpr int AS_FLAGS = AS_TOTAL, AS2_TOTAL = AS_FLAGS + 8;     // two-wave form: [0] steps begun by the eliminator, [1] rows committed by the assembler, [2] J tiles put in place by the eliminator
static_assert(AS_LJ % 2 == 0 && AS2_TOTAL * 8 <= 40960, "four assembling sweeps per CU (one per SIMD)");
// chunk forward sweep with a spike follower (k_chunk_forward): the panel of step k (43 rows x 15, then a
// zero cell and a write sink) stays in a 4-slot LDS ring for the second wave; two hand-shake cells follow
constexpr int RING_SLOT = 664;
constexpr int S_PROG = S_P + 4 * RING_SLOT;   // panels completed by the sweep
constexpr int S_CONS = S_PROG + 1;            // panels consumed by the follower
constexpr int S_RING_OUT = S_PROG + 2;        // != 0: a wait on the ring ran out (the chunk's solve is reported failed; later waits do not wait)
constexpr int RING_SPIN_MAX = 1 << 22;        // polls (each an LDS read + s_sleep): seconds, against the microseconds a step takes
constexpr int S_BC_RING = S_PROG + 8;
constexpr int S_TOTAL_RING = S_BC_RING + 16;
#ifndef VF_ASM2_ROLES
#define VF_ASM2_ROLES 1     // two-wave assembling sweep: 1 = the workgroups of a CU agree on one eliminator per SIMD (View::place), 0 = wave 0 eliminates, 2 = wave 1 does (test build)
#endif
#ifndef VF_ASM2_BSLEEP
#define VF_ASM2_BSLEEP 1    // s_sleep argument of the assembler's polls (units of 64 clocks)
#endif
#ifndef VF_ASM2_PRIO
#define VF_ASM2_PRIO 1      // two-wave assembling sweep: 1 = the eliminator's instructions issue first (s_setprio), 2 = the assembler's, 0 = neither
#endif
#ifndef VF_PIVOT_PERMLANE
#define VF_PIVOT_PERMLANE 0      // 1: the scaled pivot column is replicated over the 16-lane rows by VALU lane swaps instead of through LDS
#endif
#define WSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define VF_PIVOT_SLOT(i) do {} while (0)     // places the generated pivot code leaves for its caller (tools/gen_pivot.py)
#ifndef VF_AS_PLACEMENT
#define VF_AS_PLACEMENT 0   // which of the pivot code's 28 places take the assembling sweep's 15 pieces (as_piece_of_slot)
#endif
// piece (the I of as_piece: 0 = operand reads, 2 .. 17 = one matrix instruction each, with gaps) issued at place `slot` of the
// pivot code; 1 = none.  Placement 0: as early as possible (places 0 .. 17); 1: every other place; 2: as late as possible
constexpr int as_piece_of_slot(int slot) {
    [[maybe_unused]] constexpr int pieces[15] = {0, 2, 3, 4, 5, 7, 8, 9, 10, 12, 13, 14, 15, 16, 17};
#if VF_AS_PLACEMENT == 0
    return slot;
#elif VF_AS_PLACEMENT == 1
    return slot == 0 ? 0 : (slot % 2 == 1 ? pieces[(slot + 1) / 2] : 1);
#else
    return slot == 0 ? 0 : ((slot >= 12 && slot <= 25) ? pieces[slot - 11] : 1);
#endif
}
#ifndef VF_ASM2_CHUNK
#define VF_ASM2_CHUNK 1024
#endif
#ifndef VF_AS_SLOTS
#define VF_AS_SLOTS 1     // assembling sweep: 1 = its matrix-core pieces ride in the pivot code's places, 0 = in front of the Schur update
#endif
#define VF_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef VF_SOLVE_STAMPS   // diagnostic build only (tools/build_stamps.sh); never in the shipped library
__device__ unsigned long long g_stamps[16];
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long _t; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); __builtin_amdgcn_sched_barrier(0); if (w == 0) st[i] += _t - tprev; tprev = _t; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
// uncached read of an LDS cell another wave of the workgroup writes (hand-shake counters)
VF_DI double lds_peek(const double* p) {
    double x;
    const unsigned addr = (unsigned)(size_t)p;    // low half of a flat LDS address = LDS offset
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr) : "memory");
    return x;
}

// TW = false: one wave sweeps the whole window (throughput form, one window per SIMD).
// TW = true : burn-at-both-ends (twisted) factorisation for one-window latency: wave 0 eliminates
//   keyframes 0 .. t-1 forwards, wave 1 eliminates n-1 .. t+3 backwards (the same code on the
//   reversed sequence: block d of reversed row j is H[j+d][d]^T), they meet at a dense 45x45 system
//   for keyframes t, t+1, t+2, and the two back substitutions run concurrently.  No extra flops.
constexpr int MID_LD = 47;                      // 45 columns + rhs + pad
constexpr int MID_TOTAL = 45 * MID_LD + 48;     // + the 45 solved increments handed to both waves
// MODE 2 / 3 (partitioned solve, K4p): the forward / backward sweep of ONE chunk of a window.  The
//   forward sweep eliminates the chunk's `ni` interior keyframes and leaves the partially reduced
//   45x45 block (+ rhs) of the 3 separator keyframes that follow in `sep_out`; couplings to
//   keyframes in front of the chunk (the left separator) are dropped here and carried by the spike
//   kernel.  The backward sweep starts from the separator increments already in v.delta.
// MODE 4 / 5 (split sweep, batches of >= View::split_min windows): the forward sweep and the back substitution of SOLVE_FULL
//   as two kernels.  They hand over through HBM only (the panels and the rhs row the forward sweep stores anyway); the
//   back substitution needs 9 KB of LDS instead of 38 and so runs two and more waves per SIMD where the fused kernel
//   is held to one by the forward sweep's trailing window.
// MODE 6 (assembling forward sweep, batches of >= View::asm_min windows): SOLVE_FULL_FWD that forms the block rows of H itself,
//   from the J stream and the between linearisations K1 / K2 leave, on the matrix cores, under the waits of the pivot
//   chain -- K3 is not launched, H is neither written nor read.  See "assembling sweep" below.
// MODE 7 / 8 (k_band_forward_asm2): MODE 6 as TWO waves of one workgroup sharing its LDS -- wave 0 eliminates (SOLVE_ASM_A: the
//   forward sweep of MODE 4 whose rows somebody else commits), wave 1 assembles (SOLVE_ASM_B: the row recurrence of MODE 6 and
//   nothing else) and writes row k + 4 into the window between the eliminator's read of pivot row k and its Schur update's
//   accumulator reads; two cells of LDS carry the hand-shake.  Eight waves per CU on the LDS of four sweeps: the matrix
//   instructions of the two roles overlap each other's waits, which one wave cannot do for itself.
// MODE 9 / 10 (incremental updates, View::inc_*): the forward sweep of SOLVE_FULL started at keyframe cg.i0 of the window from a
//   checkpoint of its trailing window (and leaving one at every CK-th keyframe slot it passes), and the back substitution of
//   SOLVE_FULL_BWD that stops once it reproduces the increments that are there, below keyframe cg.ni of the window.
enum { SOLVE_FULL = 0, SOLVE_TWISTED = 1, SOLVE_CHUNK_FWD = 2, SOLVE_CHUNK_BWD = 3, SOLVE_FULL_FWD = 4, SOLVE_FULL_BWD = 5, SOLVE_ASM_FWD = 6,
       SOLVE_ASM_A = 7, SOLVE_ASM_B = 8, SOLVE_INC_FWD = 9, SOLVE_INC_BWD = 10 };
template <int MODE>
__device__ __forceinline__ void band_solve_body(const View& v, double* __restrict__ S, double* __restrict__ S_other,
                                                double* __restrict__ MID, const int w, const int lane, const int wave,
                                                const ChunkGeom cg = ChunkGeom{0, 0, 0}, double* __restrict__ sep_out = nullptr) {
    constexpr bool TW = MODE == SOLVE_TWISTED;
    constexpr bool CH = MODE == SOLVE_CHUNK_FWD || MODE == SOLVE_CHUNK_BWD;
    constexpr bool ASA = MODE == SOLVE_ASM_A, ASB = MODE == SOLVE_ASM_B;   // the two roles of the two-wave assembling sweep
    constexpr bool AS = MODE == SOLVE_ASM_FWD || ASB;    // rows of H assembled here (no K3)
    constexpr bool CW = MODE == SOLVE_FULL_FWD || AS || ASA;   // compact trailing window ("CW_" map above); the names below shadow the full map
    constexpr bool INCF = MODE == SOLVE_INC_FWD, INCB = MODE == SOLVE_INC_BWD;
    constexpr int S_GD = CW ? CW_GD : vf::S_GD, S_DUMP = CW ? CW_DUMP : vf::S_DUMP, S_ZERO = CW ? CW_ZERO : vf::S_ZERO;
    constexpr int S_ID = CW ? CW_ID : vf::S_ID, S_P = CW ? CW_P : vf::S_P, S_BC = CW ? CW_BC : vf::S_BC;
    const int lo = v.lo[w], hi = v.hi[w];
    const int n = CH ? cg.ni + (cg.has_sep ? 3 : 0) : hi - lo - (INCF ? cg.i0 : 0);     // real rows of this sweep
    const double lam = v.lambda[w];
    const size_t base = (size_t)w * v.M + lo + ((CH || INCF) ? cg.i0 : 0);
    int failed = 0;
    // sweep geometry: sweep index kk -> window keyframe j(kk); kinds of rows: 0 real, 1 identity, 2 zero
    const bool rev = TW && wave == 1;
    const int tsp = TW ? (((n - 3) / 2) & ~3) : 0;                 // split keyframe (multiple of 4)
    const int cr = TW ? n - tsp - 3 : 0;                            // pivots of the reverse sweep
    const int qpad = TW ? ((4 - (cr & 3)) & 3) : 0;                 // identity pads in front of it
    const int npiv = CH ? cg.ni : n;                                // real pivots of a one-directional sweep
    const int cnt = !TW ? ((CH && cg.has_sep) ? cg.ni : ((npiv + 3) & ~3))
                        : (rev ? cr + qpad : tsp);                  // pivots of this sweep (multiple of 4)
    auto row_kind = [=](int kk) {
        if (!TW) return kk < n ? 0 : 1;
        if (!rev) return kk < tsp + 3 ? 0 : 1;
        if (kk < qpad) return 1;
        const int j = n - 1 - (kk - qpad);
        return j >= tsp + 3 ? 0 : (j >= tsp ? 2 : 1);
    };
    auto kf_of = [=](int kk) { return rev ? n - 1 - (kk - qpad) : kk; };   // window-local keyframe of a real row
    auto pivot_real = [=](int kk) { return !TW ? kk < npiv : (rev ? kk >= qpad : true); };

    // ---- per-lane constants.  Every LDS access below is branch-free: masked-off lanes read the
    // zero cells / write the sink, so no exec-mask juggling (and no SGPR spills) in the k loop.
    if constexpr (CW) { if (lane < 46) S[S_ZERO + lane] = lane == 16 + 14 ? 1.0 : 0.0; }     // 16 zeros, then the identity strip
    else for (int e = lane; e < 16 + 225; e += 64) S[S_ZERO + e] = (e >= 16 && (e - 16) % 16 == 0) ? 1.0 : 0.0;
    const int pd = lane < 15 ? 0 : (lane < 30 ? 1 : (lane < 36 ? 2 : 3));
    const int pa = lane < 15 ? lane : (lane < 30 ? lane - 15 : (lane < 36 ? lane - 30 : lane - 36));
    int ri_ph[4];   // LDS offset of (this lane's panel row, column 0 of the pivot slot) per phase
#pragma unroll
    for (int ph = 0; ph < 4; ph++) {
        const int s0 = ph * 15;
        if constexpr (CW) {
            const int blk = pd == 0 ? CW_D0 + h_tri(pa, 0) : (pd == 1 ? CW_D1 + pa * 15 : (pd == 2 ? CW_D2 + pa * 15 : CW_D3 + pa * 6));
            ri_ph[ph] = lane < 42 ? ((ph + pd) & 3) * CW_SLOT + blk
                      : (lane == 42 ? S_GD + s0 : (lane < 58 ? S_ID + 14 - (lane - 43) : S_ZERO));
        } else
        ri_ph[ph] = lane < 42 ? S_WD + ((((ph + pd) & 3) * 15) + pa) * LDW + s0
                  : (lane == 42 ? S_GD + s0 : (lane < 58 ? S_ID + (lane - 43) * 15 : S_ZERO));
    }
    // lanes 58..63 read 15 consecutive cells from S_ZERO: S_ZERO has 16 zeros -> a zero row
    constexpr bool RINGM = MODE == SOLVE_CHUNK_FWD;        // panels also feed the spike follower (LDS ring)
    constexpr int RSLOT = RINGM ? RING_SLOT : 0;
    const int pw_off = (lane >= 15 && lane < (RINGM ? 58 : 43)) ? S_P + (lane - 15) * 15
                                                                : (RINGM ? S_P + 646 : S_DUMP + 16);   // sub-panel -> LDS
    // (the ring's zero cells and hand-shake counters are initialised by k_chunk_forward BEFORE the two waves part:
    // the follower must never poll a counter left over in LDS by an earlier workgroup)
    const int op_zero = RINGM ? S_P + 645 : S_ZERO;
    const bool follower = RINGM && cg.i0 > 0;              // chunk 0 has no left separator, hence no spike
    // Schur write-back targets (MFMA C layout): tile t in {(0,0),(1,0),(1,1)}, register r:
    //   i = 16*Ti + (lane>>4) + 4r (trailing row, 27 = rhs), j = 16*Tj + (lane&15)
    int tgt_ph[4][12];
    // (the eliminator wave of the two-wave assembling sweep has 256 registers and keeps a J tile in 80 of them: it holds one
    // base per target and forms the four phases' addresses as base + step * ((phase + shift) & 3) when it uses them)
    int tgB[12] = {}, tgM[12] = {};
#pragma unroll
    for (int q = 0; q < 12; q++) {
        const int t = q >> 2, r = q & 3;
        const int Ti = t == 0 ? 0 : 1, Tj = t == 2 ? 1 : 0;
        const int i = 16 * Ti + (lane >> 4) + 4 * r, j = 16 * Tj + (lane & 15);
        const int rs = i < 15 ? 1 : (i < 21 ? 2 : 3), ra = i < 15 ? i : (i < 21 ? i - 15 : i - 21);
        const int cs = j < 15 ? 1 : (j < 21 ? 2 : 3), ca = j < 15 ? j : (j < 21 ? j - 15 : j - 21);
        const bool valid = i <= 27 && j <= 26 && j <= i;
#pragma unroll
        for (int ph = 0; ph < 4; ph++) {
            const int cj = (((ph + cs) & 3) * 15) + ca;
            if constexpr (CW) {
                const int e = rs - cs;   // 0, 1, 2 (rows k+1 .. k+3 against columns k+1 .. k+3)
                const int blk = e == 0 ? CW_D0 + h_tri(ra, ca) : (e == 1 ? CW_D1 + ra * 15 + ca : CW_D2 + ra * 15 + ca);
                if constexpr (ASA) {
                    tgB[q] = !valid ? S_DUMP + 32 + lane : (i == 27 ? S_GD + ca : blk);
                    tgM[q] = !valid ? 0 : (i == 27 ? 15 | (cs << 16) : CW_SLOT | (rs << 16));
                } else
                tgt_ph[ph][q] = !valid ? S_DUMP + 32 + lane : (i == 27 ? S_GD + cj : ((ph + rs) & 3) * CW_SLOT + blk);
            } else
            tgt_ph[ph][q] = !valid ? S_DUMP + 32 + lane
                                   : (i == 27 ? S_GD + cj : S_WD + ((((ph + rs) & 3) * 15) + ra) * LDW + cj);
        }
    }
    // MFMA operands: P[15 + 16*T + (lane&15)][4*q + (lane>>4)]; outside the 28x15 panel -> zero cell
    int op0[4], op1[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int kc = 4 * q + (lane >> 4);
        op0[q] = kc < 15 ? S_P + (lane & 15) * 15 + kc : op_zero;
        op1[q] = (kc < 15 && 31 + (lane & 15) <= 42) ? S_P + (16 + (lane & 15)) * 15 + kc : op_zero;
    }
    // block-row commit maps (vf_kernels.hpp "Block row of H"): per lane, where the words it loads go in a slot of the window
    //   d1: idx = lane + 64 j over the 15x15 block -> (a, c)
    int cm_off[4], cm_srcT[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int idx = lane + 64 * j, a = idx / 15, c = idx - a * 15;
        const bool in = idx < 225;
        cm_off[j] = in ? a * LDW + c : -1;
        cm_srcT[j] = in ? c * 15 + a : 0;            // transposed read for the reverse sweep
    }
    //   d0: e = lane + 64 j over the 120 entries of the lower triangle -> (a, c <= a); the diagonal gets +lambda
    int t0_off[2], t0_a[2], t0_c[2];
    double t0_lam[2], t0_one[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int e = lane + 64 * j;
        int a = 0;
        while (h_tri(a + 1, 0) <= e && a < 14) a++;
        const int c = e - h_tri(a, 0);
        const bool in = e < 120;
        t0_a[j] = in ? a : -1;
        t0_c[j] = in ? c : -2;
        t0_off[j] = in ? a * LDW + c : -1;
        t0_lam[j] = (in && a == c) ? lam : 0.0;
        t0_one[j] = (in && a == c) ? 1.0 : 0.0;
    }
    //   d2, d3: 6x6 pose x pose (lane < 36); dx: 6x9 (pose) x (velocity / bias), columns 6..14 (lane < 54)
    const int s6_off = lane < 36 ? (lane / 6) * LDW + lane % 6 : -1;
    const int s6_srcT = lane < 36 ? (lane % 6) * 6 + lane / 6 : 0;      // reverse sweep: transposed
    const int x9_off = lane < 54 ? (lane / 9) * LDW + 6 + lane % 9 : -1;
    const double mp_third = (v.mp_on[w] && n >= 3 && !rev && ((!CH && !INCF) || cg.i0 == 0)) ? 1.0 : 0.0;
    WSYNC();

    // ---- assembling sweep (AS): the block row of keyframe R is formed here instead of being read from H --------------------
    // One "iteration" per row, K3's recurrence on one wave: with the operands of IMU factor R+1 in LDS (i side = keyframe R,
    // j side = keyframe R+1; column 15 of an operand tile carries the whitened residual, so J^T r comes with J^T J)
    //     Dfin = D + Ji^T [Ji | r]                       diagonal block + gradient of row R        (D: started one iteration ago)
    //     On   = Jj^T Ji,  Dn = Jj^T [Jj | r]            coupling block and start of the diagonal of row R+1
    //     Z    = X^T X,  X = [Ja | Jb | 0 0 0 | r]       everything the between factor ending at R contributes, in ONE tile (6 rows =
    //                                                    two k-steps): Z[0:6, 0:6 | 15] = Ja^T [Ja | r] goes to the diagonal block +
    //                                                    gradient of row R - d (still in the window), Z[6:12, 6:12 | 15] = Jb^T [Jb | r]
    //                                                    to those of row R, Z[6:12, 0:6] = Jb^T Ja is the pose block of row R against R - d
    // 14 v_mfma_f64_16x16x4 per row (a single wave gets one through every ~125 cycles, so their number is the price of the
    // fusion: with the three between products formed separately there were 18); inside the elimination loop they are issued one at a time from the places the generated
    // pivot code leaves for them (VF_PIVOT_SLOT), i.e. under the waits of the chain.  The J stream is read as K3 reads it --
    // whole tiles of 8 factors, every 128-B line once -- but one tile per 8 steps: the next tile waits in registers (19
    // 16-byte words per lane) and is written to LDS when the last factor of the current one has been used.  A between
    // factor's term on its OLDER keyframe goes into that row's slot of the window (the row is at most three steps from its
    // elimination, so it is still there), which is why one staged between linearisation is enough.
    struct Asm { d4_t D, O, Dfin, Z, On, Dn; double ai[4], aj[4], xx[2]; int d, d_next; };
    struct AsmNext { d2_t tj[19]; double tr[2], blA[2], blB[2]; int aA, aB, k0_next, tiles_wanted; };
    constexpr int AS_NOWN = JS_PAIRS * JT;                       // 16-byte words of a tile
    const int as_b = (AS || ASA) ? v.sel[w] : 0;
    AsmMaps am = {};
    int as_cD[4] = {}, as_cDs[4] = {}, as_cO[4] = {}, as_c1[2] = {}, as_c2[2] = {}, as_c3[2] = {}, as_rB[2] = {}, as_rS[2] = {}, as_bB[2] = {}, as_bS[2] = {}, as_oX[2] = {};
    // (signs and the diagonal's lambda are lane properties, formed where used: the window keeps -g, and the gradient is column 15 of every tile)
    const double as_sgn = (lane & 15) == 15 ? -1.0 : 1.0;
    const int as_hx = lane < 54 ? CW_D2 + (lane / 9) * 15 + 6 + lane % 9 : -1;
    if constexpr (AS) {
        am = make_asm_maps(v, w, lo, hi, lane);
        const int ci = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int a = kq + 4 * r;
            const bool tri = a < 15 && ci <= a, grad = a < 15 && ci == 15;
            as_cD[r] = tri ? CW_D0 + h_tri(a, ci) : (grad ? S_GD + a : S_DUMP + 32 + lane);      // + PH * as_cDs
            as_cDs[r] = tri ? CW_SLOT : (grad ? 15 : 0);
            as_cO[r] = (a < 15 && ci < 15) ? CW_D1 + a * 15 + ci : -1;
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            // rows 0..5 of Z (registers 0, 1): the older keyframe's term, into that row's slot
            const int a6 = kq + 4 * r;
            const bool tri = a6 < 6 && ci <= a6, grad = a6 < 6 && ci == 15;
            as_rB[r] = tri ? CW_D0 + h_tri(a6, ci) : (grad ? S_GD + a6 : S_DUMP + 32 + lane);   // + slot * as_rS
            as_rS[r] = tri ? CW_SLOT : (grad ? 15 : 0);
            // rows 6..11 of Z (registers 1, 2): columns 0..5 = the pose block against R - d, columns 6..11 | 15 = this row's own term
            const int b6 = kq + 4 * (r + 1) - 6;
            const bool brow = b6 >= 0 && b6 < 6;
            const bool in = brow && ci < 6;
            as_c1[r] = in ? CW_D1 + b6 * 15 + ci : -1;
            as_c2[r] = in ? CW_D2 + b6 * 15 + ci : -1;
            as_c3[r] = in ? CW_D3 + b6 * 6 + ci : -1;
            const bool btri = brow && ci >= 6 && ci - 6 <= b6, bgrad = brow && ci == 15;
            as_bB[r] = btri ? CW_D0 + h_tri(b6, ci - 6) : (bgrad ? S_GD + b6 : S_DUMP + 32 + lane);   // + PH * as_bS
            as_bS[r] = btri ? CW_SLOT : (bgrad ? 15 : 0);
            // operand word of k-step r: X[4 r + kq][ci] in the staged linearisation (r: 0, Ja: 6, Jb: 42; its pad cell is zero)
            const int row = 4 * r + kq;
            as_oX[r] = row >= 6 ? BTW_OUT : (ci < 6 ? 6 + row * 6 + ci : (ci < 12 ? 42 + row * 6 + ci - 6 : (ci == 15 ? row : BTW_OUT)));
        }
        if (lane < JT) S[AS_LJ + lane * LJS + LJ_ZERO] = 0.0;
        if (lane < 2) S[AS_LB + BTW_OUT + lane] = 0.0;
    }
    const size_t as_tiles = (size_t)(v.G >> 6);
    // loads of the J tile that holds window slot k0 (a multiple of 8) and of its 8 residual vectors
    auto as_tile_fetch = [=](int k0, AsmNext& nx) {
        long gt = ((long)w * v.M + k0) >> JT_LOG;
        const long gmax = (v.G >> JT_LOG) - 1;
        gt = gt > gmax ? gmax : gt;                          // (rows past the window's end: any tile, zeroed at commit)
        const d2_t* jt = (const d2_t*)(v.imu_j + ((size_t)as_b * (size_t)(v.G >> JT_LOG) + (size_t)gt) * JT_STRIDE);
#pragma unroll
        for (int it = 0; it < 19; it++) {
            const int e = it * 64 + lane;
            nx.tj[it] = jt[e < AS_NOWN ? e : AS_NOWN - 1];
        }
        const double* rb = v.imu_r + (size_t)as_b * as_tiles * IMU_R * TILE;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int e = lane + 64 * j < JT * IMU_R ? lane + 64 * j : JT * IMU_R - 1;
            const int fac = e / IMU_R, a = e - fac * IMU_R;
            const long gf = (gt << JT_LOG) + fac;
            nx.tr[j] = rb[((size_t)(gf >> 6) * IMU_R + a) * TILE + (gf & 63)];
        }
    };
    auto as_tile_commit = [&](int k0, const AsmNext& nx) {
        // word e = 64 it + lane of the tile is pair 8 it + (lane >> 3) of factor lane & 7: one address per lane, the rest immediates
        const int fac = lane & (JT - 1), kf = k0 + fac;
        const bool keep = kf > lo && kf < hi;                    // a factor outside the window contributes zeros
        double* dst = S + AS_LJ + fac * LJS + 2 * (lane >> JT_LOG);
#pragma unroll
        for (int it = 0; it < 19; it++) {
            d2_t x = nx.tj[it];
            x.x = keep ? x.x : 0.0;
            x.y = keep ? x.y : 0.0;
            if (it < 18 || lane < AS_NOWN - 18 * 64) *(d2_t*)(dst + it * (2 * 64 / JT)) = x;
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int e = lane + 64 * j;
            const int f2 = e / IMU_R, a = e - f2 * IMU_R, kf2 = k0 + f2;
            if (e < JT * IMU_R) S[AS_LJ + f2 * LJS + LJ_R + a] = (kf2 > lo && kf2 < hi) ? nx.tr[j] : 0.0;
        }
    };
    // between linearisation of the factor ending at sweep row R (words lane, lane + 64 of 78) and its older keyframe (-1: none)
    // (the older keyframe comes back as loaded, one copy per lane: decoding it here would wait for the load on the spot)
    auto as_btw_fetch = [=](int R, double (&bl)[2], int& a_raw) {
        const int ks = lo + R;
        long gs = (long)w * v.M + ks;
        gs = gs < v.G ? gs : v.G - 1;
        a_raw = v.btw_a[gs];
        const double* bo = v.btw_out + (size_t)as_b * as_tiles * BTW_OUT * TILE + (size_t)(gs >> 6) * BTW_OUT * TILE + (gs & 63);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int f = lane + 64 * j < BTW_OUT ? lane + 64 * j : BTW_OUT - 1;
            bl[j] = bo[(size_t)f * TILE];
        }
    };
    // distance 1..3 from row R to the older keyframe of the between factor ending there; 0 = no factor
    auto as_btw_dist = [=](int R, int a_raw) {
        const int ks = lo + R;
        const int a = __builtin_amdgcn_readfirstlane(a_raw);
        return (R > 0 && ks < hi && a >= lo && a < ks) ? ks - a : 0;
    };
    auto as_btw_commit = [&](const double (&bl)[2], int d) {
        S[AS_LB + lane] = d > 0 ? bl[0] : 0.0;
        S[lane + 64 < BTW_OUT ? AS_LB + lane + 64 : S_DUMP + 32 + lane] = d > 0 ? bl[1] : 0.0;
    };
    // the matrix-core work of one iteration, cut into the pieces the pivot code's places take (piece 0: operand reads)
    auto as_piece = [&](auto i_, Asm& z, const int fimg) {
        constexpr int I = decltype(i_)::value;
        auto mf = [](double a, double b, d4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); };
        if constexpr (I == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) { z.ai[q] = S[fimg + am.offI[q]]; z.aj[q] = S[fimg + am.offJ[q]]; }
#pragma unroll
            for (int q = 0; q < 2; q++) z.xx[q] = S[AS_LB + as_oX[q]];
        }
        else if constexpr (I == 2) z.Dfin = mf(z.ai[0], z.ai[0], z.D);
        else if constexpr (I == 3) z.Dn = mf(z.aj[0], z.aj[0], (d4_t){0, 0, 0, 0});
        else if constexpr (I == 4) z.On = mf(z.aj[0], z.ai[0], (d4_t){0, 0, 0, 0});
        else if constexpr (I == 5) z.Z = mf(z.xx[0], z.xx[0], (d4_t){0, 0, 0, 0});
        else if constexpr (I == 7) z.Dfin = mf(z.ai[1], z.ai[1], z.Dfin);
        else if constexpr (I == 8) z.Dn = mf(z.aj[1], z.aj[1], z.Dn);
        else if constexpr (I == 9) z.On = mf(z.aj[1], z.ai[1], z.On);
        else if constexpr (I == 10) z.Z = mf(z.xx[1], z.xx[1], z.Z);
        else if constexpr (I == 12) z.Dfin = mf(z.ai[2], z.ai[2], z.Dfin);
        else if constexpr (I == 13) z.Dn = mf(z.aj[2], z.aj[2], z.Dn);
        else if constexpr (I == 14) z.On = mf(z.aj[2], z.ai[2], z.On);
        else if constexpr (I == 15) z.Dfin = mf(z.ai[3], z.ai[3], z.Dfin);
        else if constexpr (I == 16) z.Dn = mf(z.aj[3], z.aj[3], z.Dn);
        else if constexpr (I == 17) z.On = mf(z.aj[3], z.ai[3], z.On);
    };
    // A into the diagonal block / gradient of row R - d, which sits in slot (PH - d) & 3 of the window.  Issued after the
    // write-back of the previous step's Schur update and before this step's reads of it (LDS operations of a wave retire in order).
    auto as_rmw = [&](auto ph, const Asm& z) {
        constexpr int PH = decltype(ph)::value;
        if (z.d > 0) {
            const int sa = (PH + 4 - z.d) & 3;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int addr = as_rB[r] + sa * as_rS[r];
                S[addr] = fma(as_sgn, z.Z[r], S[addr]);
            }
        }
    };
    // row R (sweep index, R & 3 == PH) into the slot the pivot keyframe frees
    // (SPECIAL: the rows that can carry the prior / the marginal prior -- the first four, committed in front of the loop; a
    // prior on a later keyframe is added by as_late_prior.  Their code, 150 loads, stays out of the elimination loop.)
    auto as_commit = [&](auto ph, auto special_, int R, const Asm& z) {
        constexpr int PH = decltype(ph)::value;
        constexpr bool SPECIAL = decltype(special_)::value != 0;
        constexpr int sb = PH * CW_SLOT;
        const int ci = lane & 15, kq = lane >> 4;
        const int kind = row_kind(R);
        const double dg = kind == 0 ? lam : (kind == 1 ? 1.0 : 0.0);
        d4_t Dv = z.Dfin, Ov = z.O;
        double t2[2], t3[2], hxv = 0.0;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            t2[r] = z.d == 2 ? z.Z[r + 1] : 0.0;        // (rows 6..11 of Z)
            t3[r] = z.d == 3 ? z.Z[r + 1] : 0.0;
        }
        const int k = lo + R;
        const bool is_prior = SPECIAL && am.prior_key == k && kind == 0;
        const int mo = (SPECIAL && am.marg_on && kind == 0) ? R : 99;
        if (SPECIAL && (is_prior || mo < 3)) {       // (wave-uniform, the window's first rows only) the prior and the marginal prior, as K3 adds them
            const double* Pq = v.prior_out + ((size_t)as_b * v.B + w) * PRIOR_OUT;
            const double* ML = v.mp_L + (size_t)w * 729;
            const double* Mg = v.mp_out + ((size_t)as_b * v.B + w) * 28;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int a = kq + 4 * r;
                double add = 0.0;
                if (is_prior && a < 15) {
                    double sum = 0.0;
                    for (int rr = 0; rr < 15; rr++)
                        sum = fma(Pq[15 + rr * 15 + a], ci < 15 ? Pq[15 + rr * 15 + ci] : Pq[rr], sum);
                    add += sum;
                }
                if (mo == 0 && a < 15) add += ci < 15 ? ML[a * 27 + ci] : Mg[a];
                if ((mo == 1 || mo == 2) && a < 6 && (ci < 6 || ci == 15)) {
                    const int ob = mo == 1 ? 15 : 21;
                    add += ci < 6 ? ML[(ob + a) * 27 + ob + ci] : Mg[ob + a];
                }
                Dv[r] += add;
                if (mo == 1 && a < 6 && ci < 15) Ov[r] += ML[(15 + a) * 27 + ci];                 // (lo+1 pose) x (lo: 15)
                if (mo == 2 && a < 6 && ci < 6) Ov[r] += ML[(21 + a) * 27 + 15 + ci];             // (lo+2 pose) x (lo+1 pose)
            }
            if (mo == 2) {
#pragma unroll
                for (int r = 0; r < 2; r++) { const int b6 = kq + 4 * (r + 1) - 6; if (b6 >= 0 && b6 < 6 && ci < 6) t2[r] += ML[(21 + b6) * 27 + ci]; }
                if (lane < 54) hxv = ML[(21 + lane / 9) * 27 + 6 + lane % 9];                       // (lo+2 pose) x (velocity / bias of lo)
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) S[as_cD[r] + PH * as_cDs[r]] = fma(as_sgn, Dv[r], (lane & 15) == (lane >> 4) + 4 * r ? dg : 0.0);
#pragma unroll
        for (int r = 0; r < 4; r++) S[as_cO[r] >= 0 ? sb + as_cO[r] : S_DUMP + 32 + lane] = Ov[r];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            S[as_c2[r] >= 0 ? sb + as_c2[r] : S_DUMP + 32 + lane] = t2[r];
            S[as_c3[r] >= 0 ? sb + as_c3[r] : S_DUMP + 32 + lane] = t3[r];
        }
        S[as_hx >= 0 ? sb + as_hx : S_DUMP + 32 + lane] = hxv;
        // the between factor's own-row term, and its pose block when it reaches back one keyframe only (that block is part of
        // the 15 x 15 coupling written above): read-modify-write behind the stores (LDS operations retire in order)
        if (z.d > 0) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int addr = as_bB[r] + PH * as_bS[r];
                S[addr] = fma(as_sgn, z.Z[r + 1], S[addr]);
            }
            if (z.d == 1) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int addr = as_c1[r] >= 0 ? sb + as_c1[r] : S_DUMP + 32 + lane;
                    S[addr] = S[addr] + z.Z[r + 1];
                }
            }
        }
    };
    // a prior on a keyframe beyond the first four rows: J^T [J | r] is formed in front of the loop (as_lp: this lane's four
    // words of the tile), and added to the row when it has just been committed (slot R & 3), before any step reads it
    double as_lp[4] = {0.0, 0.0, 0.0, 0.0};
    const int as_lp_row = (AS && am.prior_key >= lo + 4 && am.prior_key < hi) ? am.prior_key - lo : -1;
    if constexpr (AS) {
        if (as_lp_row >= 0) {
            const int ci = lane & 15, kq = lane >> 4;
            const double* Pq = v.prior_out + ((size_t)as_b * v.B + w) * PRIOR_OUT;
#pragma unroll 1
            for (int rr = 0; rr < 15; rr++) {
                const double cv = ci < 15 ? Pq[15 + rr * 15 + ci] : Pq[rr];
#pragma unroll
                for (int r = 0; r < 4; r++) { const int a = kq + 4 * r; as_lp[r] = fma(a < 15 ? Pq[15 + rr * 15 + a] : 0.0, cv, as_lp[r]); }
            }
        }
    }
    auto as_late_prior = [&](int R) {
        if (R == as_lp_row) {
            const int sl = R & 3;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int addr = as_cD[r] + sl * as_cDs[r];
                S[addr] = fma(as_sgn, as_lp[r], S[addr]);
            }
        }
    };
    // Once the operands of iteration R are in registers (piece 0), the between linearisation of row R+1 may take their place
    // in LDS: its loads are two steps old, and the newest stores in front of them in the memory queue -- the panel of
    // the previous step -- a whole step, so the wait this needs is short (at the end of the step it would stand behind
    // the panel stores just issued: 2 300 cycles per step, measured).  The loads of row R+3 go out behind it.
    auto as_stage_btw = [&](int R, Asm& z, AsmNext& nx) {
        z.d_next = as_btw_dist(R + 1, nx.aA);
        as_btw_commit(nx.blA, z.d_next);
        nx.blA[0] = nx.blB[0]; nx.blA[1] = nx.blB[1]; nx.aA = nx.aB;
        as_btw_fetch(R + 3, nx.blB, nx.aB);
    };
    auto as_advance = [&](int R, Asm& z, AsmNext& nx) {
        z.D = z.Dn;
        z.O = z.On;
        z.d = z.d_next;
        if (((lo + R + 2) & (JT - 1)) == 0) {               // factor R+2 opens a new tile: the old one has been used up
            if constexpr (ASB) {
                // two-wave form: the NEXT tile waits in the registers of the eliminator wave (which has a hundred to spare; this
                // wave has none), and that wave puts it in place once it has seen row R committed -- i.e. the operands of
                // factor R+1, the last of the old tile, read.  The first four rows, committed before the eliminator starts,
                // fetch for themselves and wait for the memory.
                if (R < 4) {
                    as_tile_fetch(nx.k0_next, nx);
                    as_tile_commit(nx.k0_next, nx);
                    nx.k0_next += JT;
                } else {
                    nx.tiles_wanted++;
                    int spin = 0;
                    while (lds_peek(S + AS_FLAGS + 2) < (double)nx.tiles_wanted && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep(VF_ASM2_BSLEEP);
                    if (spin >= (1 << 22) && lane == 0) S[AS_FLAGS + 3] = 1.0;
                }
            } else {
            as_tile_commit(nx.k0_next, nx);
            nx.k0_next += JT;
            as_tile_fetch(nx.k0_next, nx);
            }
        }
    };

    // ---- H block row prefetch (HBM -> registers) and commit (registers -> LDS slot) ---------
    // (passed by value: captured-by-reference scalars ended up in scratch memory)
    struct HRow { double h0[2], h1[4], h2, h3, hx, hg; };
    const size_t hbase = base;      // the buffer of the window's current normal equations
    const double* __restrict__ Hbase = v.H + hbase * HROW;
    const double* __restrict__ gbase = v.gvec + hbase * 15;
    const double* __restrict__ zrow = v.zrow;
    double* __restrict__ Lbase = v.Lp + base * PANEL;
    constexpr size_t Lstride = PANEL;                 // doubles between consecutive keyframes of a window
    double* __restrict__ dbase = v.delta + base * 15;
    auto fetch_row = [=](int kk) {
        HRow r;
        const bool real = row_kind(kk) == 0;
        if (!rev) {
            // unconditional loads: a block that is absent (row outside the window, or reaching in front of it) is read
            // from a row of zeros -- wave-uniform pointer selects instead of per-lane predicates (exec juggling);
            // lanes beyond a block's extent read neighbouring words that commit_row sends to the write sink
            const double* Hk = real ? Hbase + (size_t)kk * HROW : zrow;
            const double* H1 = (real && kk >= 1) ? Hk + H_D1 : zrow;
            const double* H2 = (real && kk >= 2) ? Hk + H_D2 : zrow;
            const double* H3 = (real && kk >= 3) ? Hk + H_D3 : zrow;
            const double* HX = (real && kk == 2 && mp_third > 0.0) ? Hk + H_DX : zrow;   // only the marginal prior fills it
            const double* G0 = real ? gbase + (size_t)kk * 15 : zrow;
#pragma unroll
            for (int j = 0; j < 4; j++) r.h1[j] = H1[lane + 64 * j];
#pragma unroll
            for (int j = 0; j < 2; j++) r.h0[j] = Hk[H_D0 + lane + 64 * j];
            r.h2 = H2[lane];
            r.h3 = H3[lane];
            r.hx = HX[lane];
            r.hg = G0[lane];   // negated at commit (a use here would stall on vmcnt)
        } else {
            // reversed sequence: block d of row j couples j with j+d = H[j+d][d]^T (pose x pose for d >= 2).
            // Rows t, t+1, t+2 (kind 2) keep only their couplings to the reverse part (j+d >= t+3):
            // their diagonal blocks, mutual couplings and rhs belong to the forward sweep's window.
            const int kind = row_kind(kk);
            const bool has = kind == 0 || kind == 2;
            const int j = has ? n - 1 - (kk - qpad) : 0, back = kk - qpad;
            const double* Hj = Hbase + (size_t)j * HROW;
            const bool l1 = has && back >= 1 && j + 1 >= tsp + 3, l2 = has && back >= 2 && j + 2 >= tsp + 3,
                       l3 = has && back >= 3 && j + 3 >= tsp + 3;
#pragma unroll
            for (int jj = 0; jj < 2; jj++) r.h0[jj] = (real && lane + 64 * jj < 120) ? Hj[H_D0 + lane + 64 * jj] : 0.0;   // symmetric: as stored
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const bool in = lane + 64 * jj < 225;
                r.h1[jj] = (in && l1) ? Hj[HROW + H_D1 + cm_srcT[jj]] : 0.0;
            }
            r.h2 = (lane < 36 && l2) ? Hj[2 * HROW + H_D2 + s6_srcT] : 0.0;
            r.h3 = (lane < 36 && l3) ? Hj[3 * HROW + H_D3 + s6_srcT] : 0.0;
            r.hx = 0.0;
            r.hg = (real && lane < 15) ? gbase[(size_t)j * 15 + lane] : 0.0;
        }
        return r;
    };
    // Chunk sweeps: rows at a chunk boundary keep only the dof that belong to this chunk (vf_kernels.hpp "SEP").
    //   head (chunks c >= 1): sweep rows 0, 1 are keyframes whose pose dof sit in the LEFT separator: their pose rows
    //     and columns are pinned (identity), also in the blocks of rows 1..4 that reach back to them;
    //   tail (a separator follows): rows ni+1, ni+2 contribute their pose dof only (their velocity / bias dof are the
    //     head of the next chunk).
    // Applied when the row is committed -- never on the freshly prefetched values (that would stall on vmcnt).
    auto mask_boundary_row = [=](HRow r, int kk) {
        const bool head = CH && cg.i0 > 0 && kk <= 4, tail = CH && cg.has_sep && kk > cg.ni && kk <= cg.ni + 2;
#pragma unroll
        for (int j = 0; j < 2; j++) {            // diagonal block (lower triangle)
            const int a = t0_a[j], c = t0_c[j];
            const double ident = a == c ? 1.0 : 0.0;
            if (head && kk <= 1) r.h0[j] = (a >= 6 && c >= 6) ? r.h0[j] : ident;
            if (tail) r.h0[j] = (a < 6 && c < 6) ? r.h0[j] : ident;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {            // coupling to the keyframe in front
            const int idx = lane + 64 * j, a = idx / 15, c = idx - a * 15;
            if (head) {
                if (kk == 1) r.h1[j] = (a >= 6 && c >= 6) ? r.h1[j] : 0.0;
                if (kk == 2) r.h1[j] = c >= 6 ? r.h1[j] : 0.0;
            }
            if (tail) r.h1[j] = (kk == cg.ni + 1 ? a < 6 : (a < 6 && c < 6)) ? r.h1[j] : 0.0;
        }
        if (head) {
            if (kk == 2 || kk == 3) { r.h2 = 0.0; r.hx = 0.0; }   // pose x pose coupling to a pinned keyframe: carried by the spike
            if (kk == 3 || kk == 4) r.h3 = 0.0;
            if (kk <= 1 && lane < 6) r.hg = 0.0;
        }
        if (tail && lane >= 6) r.hg = 0.0;
        return r;
    };
    auto commit_row = [&](auto ph, const HRow r_in, int kind, int kk) {   // sweep row kk with kk & 3 == PH
        constexpr int PH = decltype(ph)::value;
        constexpr int s = PH * 15, c1 = ((PH + 3) & 3) * 15, c2 = ((PH + 2) & 3) * 15, c3 = ((PH + 1) & 3) * 15;
        HRow r = r_in;
        if constexpr (CH) {
            if ((cg.i0 > 0 && kk <= 4) || (cg.has_sep && kk > cg.ni && kk <= cg.ni + 2)) r = mask_boundary_row(r_in, kk);
        }
        if constexpr (CW) {
            // the words of a block row land in their blocks in the order they were loaded: no per-lane maps
            constexpr int base = PH * CW_SLOT;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const double d0 = r.h0[j] + (kind == 0 ? t0_lam[j] : (kind == 1 ? t0_one[j] : 0.0));
                S[lane + 64 * j < 120 ? base + CW_D0 + lane + 64 * j : S_DUMP + 32 + lane] = d0;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) S[lane + 64 * j < 225 ? base + CW_D1 + lane + 64 * j : S_DUMP + 32 + lane] = r.h1[j];
            S[lane < 36 ? base + CW_D2 + (lane / 6) * 15 + lane % 6 : S_DUMP + 32 + lane] = r.h2;
            S[lane < 54 ? base + CW_D2 + (lane / 9) * 15 + 6 + lane % 9 : S_DUMP + 32 + lane] = r.hx;
            S[lane < 36 ? base + CW_D3 + lane : S_DUMP + 32 + lane] = r.h3;
            S[lane < 15 ? S_GD + s + lane : S_DUMP + 32 + lane] = -r.hg;
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const double d0 = r.h0[j] + (kind == 0 ? t0_lam[j] : (kind == 1 ? t0_one[j] : 0.0));
            S[t0_off[j] >= 0 ? S_WD + s * LDW + s + t0_off[j] : S_DUMP + 32 + lane] = d0;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) S[cm_off[j] >= 0 ? S_WD + s * LDW + c1 + cm_off[j] : S_DUMP + 32 + lane] = r.h1[j];
        // pose rows against the keyframes two and three back: the 6x6 blocks, and columns 6..14 (zero, but for the
        // marginal prior's entries in the window's third row): the panel reads whole rows
        S[s6_off >= 0 ? S_WD + s * LDW + c2 + s6_off : S_DUMP + 32 + lane] = r.h2;
        S[x9_off >= 0 ? S_WD + s * LDW + c2 + x9_off : S_DUMP + 32 + lane] = r.hx;
        S[s6_off >= 0 ? S_WD + s * LDW + c3 + s6_off : S_DUMP + 32 + lane] = r.h3;
        S[x9_off >= 0 ? S_WD + s * LDW + c3 + x9_off : S_DUMP + 32 + lane] = 0.0;
        S[lane < 15 ? S_GD + s + lane : S_DUMP + 32 + lane] = -r.hg;
    };
    const int n4 = cnt;   // pivots of this sweep; identity rows beyond the real ones are eliminated harmlessly
    // byte offsets of this lane's panel entries within a keyframe's panel (forward sweep: lane 15 + r holds row r)
    unsigned pp_off[8];
#pragma unroll
    for (int c = 0; c < 7; c++) pp_off[c] = (unsigned)panel_pair_off(lane >= 15 && lane < 58 ? lane - 15 : 0, c);
    pp_off[7] = (unsigned)((PANEL_LAST + (lane >= 15 && lane < 58 ? lane - 15 : 0)) * sizeof(double));
#ifdef VF_SOLVE_STAMPS
    unsigned long long st[16] = {0}, tprev = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (MODE != SOLVE_CHUNK_BWD && MODE != SOLVE_FULL_BWD && !INCB) {
    Asm az;
    AsmNext anx;
    if constexpr (AS) {
        az.D = az.O = az.Dfin = az.Z = az.On = az.Dn = (d4_t){0, 0, 0, 0};     // row 0 has no factor in front of it
        az.d = az.d_next = 0;
        anx.k0_next = (lo + 1) & ~(JT - 1);
        anx.tiles_wanted = 0;
        as_tile_fetch(anx.k0_next, anx);
        as_tile_commit(anx.k0_next, anx);
        anx.k0_next += JT;
        if constexpr (!ASB) as_tile_fetch(anx.k0_next, anx);
        { const double z2[2] = {0.0, 0.0}; as_btw_commit(z2, 0); }
        as_btw_fetch(1, anx.blA, anx.aA);
        as_btw_fetch(2, anx.blB, anx.aB);
        WSYNC();
        // one iteration outside the elimination loop: the first four rows (whatever the form), and every row of the assembler wave
        auto whole_row = [&](auto ph, auto special_, int R) {
            const int fimg = AS_LJ + ((lo + R + 1) & (JT - 1)) * LJS;
            as_piece(IC<0>{}, az, fimg);
            as_stage_btw(R, az, anx);
            as_piece(IC<2>{}, az, fimg);  as_piece(IC<3>{}, az, fimg);  as_piece(IC<4>{}, az, fimg);  as_piece(IC<5>{}, az, fimg);
            as_piece(IC<7>{}, az, fimg);  as_piece(IC<8>{}, az, fimg);  as_piece(IC<9>{}, az, fimg);
            as_piece(IC<10>{}, az, fimg); as_piece(IC<12>{}, az, fimg); as_piece(IC<13>{}, az, fimg);
            as_piece(IC<14>{}, az, fimg); as_piece(IC<15>{}, az, fimg); as_piece(IC<16>{}, az, fimg); as_piece(IC<17>{}, az, fimg);
            if constexpr (ASB && decltype(special_)::value == 0) {
                // the window may be touched once the eliminator has read pivot row R - 4 (its Schur write-back of the step
                // before is then done as well) and until it reads the accumulators of that step -- it waits for us there
                int spin = 0;
                while (lds_peek(S + AS_FLAGS) < (double)(R - 4) && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep(VF_ASM2_BSLEEP);
                if (spin >= (1 << 22) && lane == 0) S[AS_FLAGS + 3] = 1.0;      // (never seen: the eliminator reports it as a failed solve)
            }
            as_rmw(ph, az);
            as_commit(ph, special_, R, az);
            if constexpr (decltype(special_)::value == 0) as_late_prior(R);
            WSYNC();
            if constexpr (ASB) { if (lane == 0) S[AS_FLAGS + 1] = (double)R; }       // rows <= R are in the window
            as_advance(R, az, anx);
            WSYNC();
        };
        whole_row(IC<0>{}, IC<1>{}, 0);
        whole_row(IC<1>{}, IC<1>{}, 1);
        whole_row(IC<2>{}, IC<1>{}, 2);
        whole_row(IC<3>{}, IC<1>{}, 3);
        if constexpr (ASB) {
#pragma unroll 1
            for (int R = 4; R < cnt + 4; R += 4) {
                whole_row(IC<0>{}, IC<0>{}, R);
                whole_row(IC<1>{}, IC<0>{}, R + 1);
                whole_row(IC<2>{}, IC<0>{}, R + 2);
                whole_row(IC<3>{}, IC<0>{}, R + 3);
            }
            return;
        }
    } else if constexpr (ASA) {
        // the tile behind the one iteration 4 works on (factor lo + 5): ours to prefetch and to put in place from now on
        anx.k0_next = (((lo + 5) >> JT_LOG) + 1) << JT_LOG;
        anx.tiles_wanted = 0;
        as_tile_fetch(anx.k0_next, anx);
        int spin = 0;                  // the assembler wave commits the first four rows
        while (lds_peek(S + AS_FLAGS + 1) < 3.0 && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        if (spin >= (1 << 22)) failed = 1;
    } else {
    commit_row(IC<0>{}, fetch_row(0), row_kind(0), 0);
    commit_row(IC<1>{}, fetch_row(1), row_kind(1), 1);
    commit_row(IC<2>{}, fetch_row(2), row_kind(2), 2);
    commit_row(IC<3>{}, fetch_row(3), row_kind(3), 3);
    if constexpr (INCF) {
        // the sweep's trailing window as the elimination of the keyframes in front of keyframe cg.i0 left it: the 27 dof
        // [k: 15][k+1: pose][k+2: pose] and their rhs, over the rows just committed (slots 0 .. 2: sweep index 0 is phase 0)
        if (cg.i0 > 0) {
            const double* __restrict__ ckp = v.ck + (base >> CK_LOG) * CK_SZ;
            for (int e = lane; e < SEP * 28; e += 64) {
                const int i = e / 28, jc = e - i * 28;
                const int oi = i < 15 ? 0 : (i < 21 ? 1 : 2), ai = i < 15 ? i : (i < 21 ? i - 15 : i - 21);
                const int oj = jc < 15 ? 0 : (jc < 21 ? 1 : 2), aj = jc < 15 ? jc : (jc < 21 ? jc - 15 : jc - 21);
                const double val = ckp[e];
                if (jc == 27) S[S_GD + oi * 15 + ai] = val;
                else if (jc <= i) S[S_WD + (oi * 15 + ai) * LDW + oj * 15 + aj] = val;
            }
        }
    }
    }
    WSYNC();
    // ---- one elimination step, phase PH = k & 3 compile-time --------------------------------
    // `pend`, `pend2` = block rows of keyframes k+4, k+5 (fetched two steps and one step ago); this step fetches k+6.
    auto step = [&](auto ph, int k, HRow& pend, HRow& pend2) {
        constexpr int PH = decltype(ph)::value;
        STAMP(0);
        if constexpr (INCF) {
            // a checkpoint at every CK-th keyframe SLOT (sweeps of later updates start at other keyframes of the window, the
            // slot grid stays): keyframe k + j sits in slot (PH + j) & 3 of the window
            if (k > 0 && ((base + (size_t)k) & (CK - 1)) == 0) {
                double* __restrict__ ckp = v.ck + ((base + (size_t)k) >> CK_LOG) * CK_SZ;
                for (int e = lane; e < SEP * 28; e += 64) {
                    const int i = e / 28, jc = e - i * 28;
                    const int oi = i < 15 ? 0 : (i < 21 ? 1 : 2), ai = i < 15 ? i : (i < 21 ? i - 15 : i - 21);
                    double val;
                    if (jc == 27) val = S[S_GD + ((PH + oi) & 3) * 15 + ai];
                    else {
                        const int hi_i = i >= jc ? i : jc, lo_i = i >= jc ? jc : i;
                        const int oa = hi_i < 15 ? 0 : (hi_i < 21 ? 1 : 2), a = hi_i < 15 ? hi_i : (hi_i < 21 ? hi_i - 15 : hi_i - 21);
                        const int ob = lo_i < 15 ? 0 : (lo_i < 21 ? 1 : 2), bb = lo_i < 15 ? lo_i : (lo_i < 21 ? lo_i - 15 : lo_i - 21);
                        val = S[S_WD + (((PH + oa) & 3) * 15 + a) * LDW + ((PH + ob) & 3) * 15 + bb];
                    }
                    ckp[e] = val;
                }
            }
        }
        double p[15];
#pragma unroll
        for (int c = 0; c < 15; c++) p[c] = S[ri_ph[PH] + c];
        if constexpr (CW) {     // rows of keyframe k+3 (lanes 36..41) hold pose columns only: the 6 x 6 block has no columns 6..14
            const bool narrow = lane >= 36 && lane < 42;
#pragma unroll
            for (int c = 6; c < 15; c++) p[c] = narrow ? 0.0 : p[c];
        }
        if constexpr (ASA) {    // pivot row k is in registers (and the write-back of step k - 1 behind us): its slot and the rows of the window are the assembler's
            WSYNC();
            if (lane == 0) S[AS_FLAGS] = (double)k;
        }
        STAMP(1);
        // panel factorisation: straight-line code in a fixed issue order (tools/gen_pivot.py); a non-positive
        // pivot turns the last reciprocal into NaN / inf, tested once per step
        double pv_inv;
        const int pv_bcw = lane < 15 ? (RINGM ? S_BC_RING : S_BC) + lane : S_DUMP + 32 + lane;
        const int pv_bcr = (RINGM ? S_BC_RING : S_BC) + (lane & 15);
        // (assembling sweep: iteration k + 4 -- row k + 4 from the operands of factor k + 5 -- rides in the pivot code's places)
        const int as_fimg = AS_LJ + ((lo + k + 5) & (JT - 1)) * LJS;
#undef VF_PIVOT_SLOT
#if VF_AS_SLOTS
#define VF_PIVOT_SLOT(i) do { if constexpr (AS) { as_piece(IC<as_piece_of_slot(i)>{}, az, as_fimg); if constexpr ((i) == 1) as_stage_btw(k + 4, az, anx); if constexpr ((i) == 26) as_rmw(ph, az); VF_SB(); } } while (0)
#else
#define VF_PIVOT_SLOT(i) do {} while (0)
#endif
#if VF_PIVOT_PERMLANE
#include "vf_pivot_15p.inc"
        (void)pv_bcw; (void)pv_bcr;
#else
#include "vf_pivot_15.inc"
#endif
#undef VF_PIVOT_SLOT
#define VF_PIVOT_SLOT(i) do {} while (0)
        (void)as_fimg;
        if (!(pv_inv < 1e300)) failed = 1;
        // (two-wave form: ask now how far the assembler is -- the answer travels under the panel's stores, and if row k + 4
        // is there already, which is the rule, the wait in front of the Schur update costs no LDS round trip)
        double as_seen = 0.0;
        if constexpr (ASA) as_seen = *(volatile double*)(S + AS_FLAGS + 1);
        STAMP(2);
        // sub-panel + rhs -> LDS (MFMA operands); rows 15..57 -> HBM, one 128-B line per lane
        if constexpr (RINGM) {
            // slot PH still holds the panel of step k-4: wait until the follower has read it
            // (bounded, like every hand-shake of the two-wave sweep: a follower that never reports must end as a failed solve,
            // not as a wave that spins until the GPU is reset)
            if (follower) {
                int spin = 0;
                while (lds_peek(S + S_CONS) < (double)(k - 3) && lds_peek(S + S_RING_OUT) == 0.0 && ++spin < RING_SPIN_MAX) __builtin_amdgcn_s_sleep(1);
                if (spin >= RING_SPIN_MAX) { failed = 1; if (lane == 0) S[S_RING_OUT] = 1.0; }
            }
        }
#pragma unroll
        for (int c = 0; c < 15; c++) S[pw_off + PH * RSLOT + c] = p[c];
        if (lane >= 15 && lane < 58 && pivot_real(k)) {
            VF_GLOBAL char* Lk = uniform_gptr(Lbase + (size_t)kf_of(k) * Lstride);     // (the lane's place in each pair: pp_off)
#pragma unroll
            for (int c = 0; c < 7; c++) { d2_t t; t.x = p[2 * c]; t.y = p[2 * c + 1]; *(VF_GLOBAL d2_t*)(Lk + pp_off[c]) = t; }
            *(VF_GLOBAL double*)(Lk + pp_off[7]) = p[14];
        }
        WSYNC();
        if constexpr (RINGM) { if (lane == 0) S[S_PROG] = (double)(k + 1); }   // panel k is complete in its ring slot
        STAMP(3);
#if !VF_AS_SLOTS
        if constexpr (AS) {
            // the iteration's 18 matrix-core instructions, back to back: nothing else of this wave wants the vector unit
            // here, the staging of the next between linearisation and the loads behind it issue in between
            as_piece(IC<0>{}, az, as_fimg);
            as_piece(IC<2>{}, az, as_fimg);  as_piece(IC<3>{}, az, as_fimg);  as_piece(IC<4>{}, az, as_fimg);
            as_stage_btw(k + 4, az, anx);
            as_piece(IC<5>{}, az, as_fimg);  as_piece(IC<7>{}, az, as_fimg);  as_piece(IC<8>{}, az, as_fimg);
            as_piece(IC<9>{}, az, as_fimg);  as_piece(IC<10>{}, az, as_fimg); as_piece(IC<12>{}, az, as_fimg);
            as_piece(IC<13>{}, az, as_fimg); as_piece(IC<14>{}, az, as_fimg); as_piece(IC<15>{}, az, as_fimg); as_piece(IC<16>{}, az, as_fimg);
            as_piece(IC<17>{}, az, as_fimg);
            as_rmw(ph, az);
        }
#endif
        if constexpr (ASA) {    // row k + 4 committed, the between terms it brings added to rows k + 1 .. k + 3
            int spin = 0;
            if (as_seen < (double)(k + 4))
                while (lds_peek(S + AS_FLAGS + 1) < (double)(k + 4) && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep(1);
#ifdef VF_RING_WITHHOLD     // fault-injection build only (tools/variants/libvilfusion_withhold.so): window 1's eliminator is told, once,
            if (w == 1 && k == 8) spin = 1 << 22;          // that its assembler never answered
#endif
            if (spin >= (1 << 22)) failed = 1;
            if (((lo + k + 6) & (JT - 1)) == 0) {      // factor k + 6 opens a new tile, and the assembler has read the last operands of the old one
                as_tile_commit(anx.k0_next, anx);
                WSYNC();
                anx.tiles_wanted++;
                if (lane == 0) S[AS_FLAGS + 2] = (double)anx.tiles_wanted;
                anx.k0_next += JT;
                as_tile_fetch(anx.k0_next, anx);
            }
        }
        // Schur update on the matrix cores: acc[t] = P_Ti P_Tj^T for the 3 lower 16x16 tiles
        // (the trailing entries are the accumulator input and the A operands are negated: T - P P^T leaves the matrix
        // cores ready to be written back, no accumulator read-out + subtraction pass)
        double a0[4], a1[4], n0[4], n1[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { a0[q] = S[op0[q] + PH * RSLOT]; a1[q] = S[op1[q] + PH * RSLOT]; }
        d4_t acc0, acc1, acc2;
        int tg[12];
#pragma unroll
        for (int q = 0; q < 12; q++) {
            if constexpr (ASA) tg[q] = tgB[q] + (tgM[q] & 0xffff) * ((PH + (tgM[q] >> 16)) & 3);
            else tg[q] = tgt_ph[PH][q];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) { acc0[r] = S[tg[r]]; acc1[r] = S[tg[4 + r]]; acc2[r] = S[tg[8 + r]]; }
#pragma unroll
        for (int q = 0; q < 4; q++) { n0[q] = -a0[q]; n1[q] = -a1[q]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(n0[q], a0[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(n1[q], a0[q], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(n1[q], a1[q], acc2, 0, 0, 0);
        }
        // While the twelve MFMAs run (64 cycles each on this part: the longest stretch of the step that needs no issue
        // slots), row k+4 is committed to the slot the pivot keyframe frees -- its panel is in registers, the operand and
        // accumulator reads above are ahead of these writes in the LDS queue, and the write-back below goes to other rows.
        if constexpr (AS) {
            as_commit(ph, IC<0>{}, k + 4, az);
            as_late_prior(k + 4);
            as_advance(k + 4, az, anx);          // (a dispatch on the phase around ONE copy of this was slower: the register copies where its four arms meet)
        } else if constexpr (ASA) {
            // (nothing to commit or to fetch: the partner wave does both)
        } else {
        commit_row(ph, pend, row_kind(k + 4), k + 4);
        pend = pend2;
        pend2 = fetch_row(k + 6);              // also in the shadow; two steps of slack for the HBM round trip
        }
        STAMP(4);
#pragma unroll
        for (int r = 0; r < 4; r++) { S[tg[r]] = acc0[r]; S[tg[4 + r]] = acc1[r]; S[tg[8 + r]] = acc2[r]; }
        WSYNC();
        STAMP(5);
    };
    {
        HRow pend = {}, pend2 = {};
        if constexpr (!AS && !ASA) { pend = fetch_row(4); pend2 = fetch_row(5); }
#pragma unroll 1
        for (int k = 0; k < n4; k += 4) {
            step(IC<0>{}, k, pend, pend2);
            step(IC<1>{}, k + 1, pend, pend2);
            step(IC<2>{}, k + 2, pend, pend2);
            step(IC<3>{}, k + 3, pend, pend2);
        }
    }
    }   // forward sweep
#ifdef VF_K4_FWD_ONLY   // probe build only (tools/build_variant.sh): the forward sweep's share of the un-stamped kernel
    if constexpr (MODE == SOLVE_FULL) return;
#endif
    if constexpr (MODE == SOLVE_FULL_FWD || AS || ASA || INCF) {
        if constexpr (ASA) { if (lds_peek(S + AS_FLAGS + 3) != 0.0) failed = 1; }     // a wait of the assembler wave ran out
        if (lane == 0) v.fail[w] = failed;
#ifdef VF_SOLVE_STAMPS
        if (w == 0 && lane == 0) { for (int i = 0; i < 6; i++) g_stamps[i] = st[i]; for (int i = 11; i < 16; i++) g_stamps[i] = st[i]; }
#endif
        return;
    }
    if constexpr (MODE == SOLVE_CHUNK_FWD) {
        // the cut keyframe and the two after it sit in slots 0..2 (cnt is a multiple of 4): the separator's 27 dof
        // (15 + pose + pose): own H + lambda + Schur terms of this chunk's interior, and rhs, go to sep_out [27][28]
        if (cg.has_sep) {
            for (int e = lane; e < SEP * 28; e += 64) {
                const int i = e / 28, jc = e - i * 28;
                const int oi = i < 15 ? 0 : (i < 21 ? 1 : 2), ai = i < 15 ? i : (i < 21 ? i - 15 : i - 21);
                double val;
                if (jc == 27) {
                    val = S[S_GD + oi * 15 + ai];
                } else {
                    const int hi_i = i >= jc ? i : jc, lo_i = i >= jc ? jc : i;
                    const int oa = hi_i < 15 ? 0 : (hi_i < 21 ? 1 : 2), a = hi_i < 15 ? hi_i : (hi_i < 21 ? hi_i - 15 : hi_i - 21);
                    const int ob = lo_i < 15 ? 0 : (lo_i < 21 ? 1 : 2), bb = lo_i < 15 ? lo_i : (lo_i < 21 ? lo_i - 15 : lo_i - 21);
                    val = S[S_WD + (oa * 15 + a) * LDW + ob * 15 + bb];
                }
                sep_out[e] = val;
            }
        }
        if (lane == 0 && failed) atomicOr(v.fail + w, 1);
        return;
    }

    // ---- back substitution: delta_k = L_kk^-T (y_k - sum_p L[p][k-cols]^T delta(p)) -----------
    // Panel rows in HBM: 0..26 sub-diagonal rows (p = 15..41), 27 = y, 28..42 = L_kk^-T.
    // Lane r < 43 holds row r in registers; rows 0..27 go through LDS for the column sums,
    // L^-T stays in the registers of lanes 28..42.  Panels are prefetched four steps ahead, each into the register
    // slot (k & 3) it is consumed from (three steps with a rotating triple measured 2.5 % slower under a full batch;
    // touching the lines further ahead with a dword load does not help: vector loads return in order).
    S[S_DL + lane] = 0.0;
    if constexpr (MODE == SOLVE_CHUNK_BWD) {
        WSYNC();
        // increments of the separator dof (solved by k_sep_solve) start the recursion; the velocity / bias dof of the
        // two keyframes after the cut belong to the next chunk and do not couple to this one
        if (cg.has_sep && lane < 45) S[S_DL + lane] = (lane < 15 || (lane % 15) < 6) ? dbase[(size_t)cg.ni * 15 + lane] : 0.0;
        WSYNC();
    }
    if constexpr (TW) {
        __syncthreads();   // both forward sweeps done; their trailing windows are in S (left) / S_other
        if (wave == 0) {
            // dense 45x45 system of keyframes t, t+1, t+2 = left window (original H of these rows +
            // left Schur terms) + right window (Schur terms only: its rows were committed as zeros,
            // reversed order => transposed blocks).  Only regions that are written are read.
            const int sR[3] = {((qpad + n - 1 - tsp) & 3) * 15, ((qpad + n - 2 - tsp) & 3) * 15, ((qpad + n - 3 - tsp) & 3) * 15};
            for (int e = lane; e < 45 * 46; e += 64) {
                const int i = e / 46, jc = e - i * 46;
                const int oi = i / 15, ai = i - oi * 15;
                double val;
                if (jc == 45) {
                    val = S[S_GD + oi * 15 + ai] + S_other[S_GD + sR[oi] + ai];
                } else {
                    const int hi_i = i >= jc ? i : jc, lo_i = i >= jc ? jc : i;     // symmetric: fill both triangles
                    const int oa = hi_i / 15, a = hi_i - oa * 15, ob = lo_i / 15, bb = lo_i - ob * 15;
                    double vl = 0.0, vr = 0.0;
                    if (oa - ob < 2 || a < 6) vl = S[S_WD + (oa * 15 + a) * LDW + ob * 15 + bb];
                    // right window: row = keyframe with the larger sweep index = the smaller keyframe (ob)
                    if (oa == ob) vr = S_other[S_WD + (sR[oa] + a) * LDW + sR[oa] + bb];
                    else if (oa - ob == 1 || bb < 6) vr = S_other[S_WD + (sR[ob] + bb) * LDW + sR[oa] + a];
                    val = vl + vr;
                }
                MID[i * MID_LD + jc] = val;
            }
            WSYNC();
            for (int c = 0; c < 45; c++) {       // Gaussian elimination (SPD: no pivoting), entry-parallel
                const double piv = MID[c * MID_LD + c];
                if (!(piv > 0.0)) failed = 1;
                const double inv = 1.0 / piv;
                const int m = 44 - c;
                for (int e = lane; e < m * (m + 1); e += 64) {
                    const int i = c + 1 + e / (m + 1), jc = c + 1 + (e - (e / (m + 1)) * (m + 1));
                    MID[i * MID_LD + jc] -= MID[i * MID_LD + c] * inv * MID[c * MID_LD + jc];
                }
                WSYNC();
            }
            for (int i = 44; i >= 0; i--) {      // back substitution on the upper triangle
                const double xi = MID[i * MID_LD + 45] / MID[i * MID_LD + i];
                if (lane < i) MID[lane * MID_LD + 45] -= MID[lane * MID_LD + i] * xi;
                if (lane == 0) MID[45 * MID_LD + i] = xi;
                WSYNC();
            }
            if (lane < 45) dbase[(size_t)tsp * 15 + lane] = MID[45 * MID_LD + lane];
        }
        __syncthreads();
        // increments of the keyframes just beyond this sweep's last pivot, in its own slot order
        if (!rev) {          // rows t (15), t+1, t+2 at sweep indices tsp, tsp+1, tsp+2 (tsp % 4 == 0)
            if (lane < 45) S[S_DL + lane] = MID[45 * MID_LD + lane];
        } else {             // sweep indices cnt, cnt+1, cnt+2 <-> keyframes t+2, t+1, t
            if (lane < 45) { const int o = lane / 15, a = lane - o * 15; S[S_DL + ((cnt + o) & 3) * 15 + a] = MID[45 * MID_LD + (2 - o) * 15 + a]; }
        }
        WSYNC();
    }
    struct PRow { d2_t x[8]; };
#ifndef VF_BWD_PD
#define VF_BWD_PD 2      // (2 against 4 slots: solve stage 3.44 against 3.48-3.50 ms at 1 024 windows, 178 against 240 registers; round 5)
#endif
    static_assert(VF_BWD_PD == 2 || VF_BWD_PD == 4, "the back substitution keeps 2 or 4 panel slots: any other depth reads panels it has not loaded");
    constexpr int PD = (MODE == SOLVE_FULL_BWD || INCB) ? VF_BWD_PD : 4;      // panels prefetched ahead of the recursion
    // Backward sweep: lane r < 28 holds panel row r (sub-diagonal rows and the rhs row: they go through LDS); the rows of
    // L^-T sit in lanes XL .. XL+14 of ONE 16-lane row, where s and x are formed as well, so that both matrix-vector
    // products of the recursion broadcast their vector with DPP row_newbcast inside v_fmac_f64 (one instruction per term
    // instead of v_readlane x2, s_nop, v_fma); every other lane points at the keyframe's zero cell.
    constexpr int XL = 32;
    const bool xl_lane = lane >= XL && lane < XL + 15;
    unsigned pb_off[8];
    {
        const int brow = lane < 28 ? lane : (xl_lane ? 28 + (lane - XL) : -1);
#pragma unroll
        for (int c = 0; c < 7; c++) pb_off[c] = brow >= 0 ? (unsigned)panel_pair_off(brow, c) : (unsigned)(PANEL_DUMP * sizeof(double));
        pb_off[7] = (unsigned)((brow >= 0 ? PANEL_LAST + brow : PANEL_DUMP) * sizeof(double));
    }
    auto load_panel = [=](int k) {   // not a real pivot: any valid panel is loaded and zeroed at use (no use here: no stall)
        PRow r;
        const bool ok = k >= 0 && k < cnt && pivot_real(k);
        const VF_GLOBAL char* Lk = uniform_gptr(Lbase + (size_t)(ok ? kf_of(k) : 0) * Lstride);
#pragma unroll
        for (int c = 0; c < 7; c++) r.x[c] = *(const VF_GLOBAL d2_t*)(Lk + pb_off[c]);
        r.x[7].x = *(const VF_GLOBAL double*)(Lk + pb_off[7]);
        r.x[7].y = 0.0;
        return r;
    };
    const int bw_off = lane < 28 ? S_P + lane * 15 : S_DUMP + 16;
    const int col = xl_lane ? lane - XL : 0;
    const int dl_w = xl_lane ? S_DL + lane - XL : S_DUMP + 32 + lane;
    WSYNC();
    // Software-pipelined: the recursion delta_{k+1} -> delta_k runs on registers only --
    //   s = y - part - sum_a P[a][.] delta_{k+1}[a]   (delta_{k+1} = the previous step's x, broadcast by DPP)
    //   x = L_kk^-T s                                  (s broadcast by DPP)
    // while everything that does not depend on delta_{k+1} is prepared one step ahead (`prep`): the panel rows of the
    // next keyframe go through LDS into per-column registers, and the couplings to the two keyframes further on
    // (12 pose columns, increments already in LDS) are summed into `part`.  The step was 3 LDS round trips and an HBM
    // wait in sequence (about 2 600 cycles for 200 instructions); LDS accesses of one wave execute in issue order, so
    // the write -> read hand-offs below need a compiler barrier, not a wait.
#define CBAR() asm volatile("" ::: "memory")
    struct Col { double row[15], pv[15], y, part, dold; };   // of one keyframe, in lanes XL .. XL+14: its L^-T row, its column data
    int inc_run = 0;      // INCB: consecutive keyframes whose increment came out as it was
    auto prep = [&](auto ph, int k, PRow& slot, Col& o) {
        constexpr int PH = decltype(ph)::value;          // = k & 3
        constexpr int b2 = S_DL + ((PH + 2) & 3) * 15, b3 = S_DL + ((PH + 3) & 3) * 15;
        STAMP(6);
        const double keep = (k >= 0 && pivot_real(k)) ? 1.0 : 0.0;   // identity rows: zero panel
#pragma unroll
        for (int c = 0; c < 7; c++) { o.row[2 * c] = keep * slot.x[c].x; o.row[2 * c + 1] = keep * slot.x[c].y; }
        o.row[14] = keep * slot.x[7].x;
        if constexpr (INCB) o.dold = (xl_lane && k >= 0 && pivot_real(k)) ? dbase[(size_t)kf_of(k) * 15 + lane - XL] : 0.0;
        slot = load_panel(k - PD);   // PD steps ahead, into the slot just consumed (slot = k & (PD - 1): no register rotation)
#pragma unroll
        for (int c = 0; c < 15; c++) S[bw_off + c] = o.row[c];
        CBAR();
        STAMP(7);
#pragma unroll
        for (int a = 0; a < 15; a++) o.pv[a] = S[S_P + a * 15 + col];
        double q[12], d[12];
#pragma unroll
        for (int a = 0; a < 12; a++) q[a] = S[S_P + (15 + a) * 15 + col];
#pragma unroll
        for (int a = 0; a < 6; a++) { d[a] = S[b2 + a]; d[6 + a] = S[b3 + a]; }
        o.y = S[S_P + 27 * 15 + col];
        double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int a = 0; a < 12; a += 3) {
            t0 = fma(q[a], d[a], t0);
            t1 = fma(q[a + 1], d[a + 1], t1);
            t2 = fma(q[a + 2], d[a + 2], t2);
        }
        o.part = (t0 + t1) + t2;
        CBAR();                      // the next prep overwrites these panel rows in LDS
        STAMP(8);
    };
    auto solve = [&](auto ph, int k, const Col& c_, double& xprev) {
        constexpr int PH = decltype(ph)::value;
        STAMP(9);
        // acc += (lane XL + N of xsrc) * other, on the lanes of that 16-lane row (the DPP operand must have been written
        // at least two wait states earlier: s_nop in front of the first use of a fresh vector)
        auto bfma = [](auto n_, double& acc, const double bsrc, const double other) {
            constexpr int N = decltype(n_)::value;
            asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(other), "n"(N));
        };
        double s0 = c_.part, s1 = 0.0, s2 = 0.0;
        asm volatile("s_nop 1" :: "v"(xprev));
        bfma(IC<0>{}, s0, xprev, c_.pv[0]);   bfma(IC<1>{}, s1, xprev, c_.pv[1]);   bfma(IC<2>{}, s2, xprev, c_.pv[2]);
        bfma(IC<3>{}, s0, xprev, c_.pv[3]);   bfma(IC<4>{}, s1, xprev, c_.pv[4]);   bfma(IC<5>{}, s2, xprev, c_.pv[5]);
        bfma(IC<6>{}, s0, xprev, c_.pv[6]);   bfma(IC<7>{}, s1, xprev, c_.pv[7]);   bfma(IC<8>{}, s2, xprev, c_.pv[8]);
        bfma(IC<9>{}, s0, xprev, c_.pv[9]);   bfma(IC<10>{}, s1, xprev, c_.pv[10]); bfma(IC<11>{}, s2, xprev, c_.pv[11]);
        bfma(IC<12>{}, s0, xprev, c_.pv[12]); bfma(IC<13>{}, s1, xprev, c_.pv[13]); bfma(IC<14>{}, s2, xprev, c_.pv[14]);
        const double s = c_.y - ((s0 + s1) + s2);
        // x = L^-T s on lanes XL .. XL+14 (row c of L^-T in registers)
        double x0 = 0.0, x1 = 0.0;
        asm volatile("s_nop 1" :: "v"(s));
        bfma(IC<0>{}, x0, s, c_.row[0]);   bfma(IC<1>{}, x1, s, c_.row[1]);   bfma(IC<2>{}, x0, s, c_.row[2]);   bfma(IC<3>{}, x1, s, c_.row[3]);
        bfma(IC<4>{}, x0, s, c_.row[4]);   bfma(IC<5>{}, x1, s, c_.row[5]);   bfma(IC<6>{}, x0, s, c_.row[6]);   bfma(IC<7>{}, x1, s, c_.row[7]);
        bfma(IC<8>{}, x0, s, c_.row[8]);   bfma(IC<9>{}, x1, s, c_.row[9]);   bfma(IC<10>{}, x0, s, c_.row[10]); bfma(IC<11>{}, x1, s, c_.row[11]);
        bfma(IC<12>{}, x0, s, c_.row[12]); bfma(IC<13>{}, x1, s, c_.row[13]); bfma(IC<14>{}, x0, s, c_.row[14]);
        const double x = x0 + x1;
        S[xl_lane ? S_DL + PH * 15 + lane - XL : dl_w] = x;
        // (head keyframes of a chunk: only their velocity / bias increments are this sweep's; the pose part is the separator's)
        if (xl_lane && pivot_real(k) && !(CH && cg.i0 > 0 && k < 2 && lane < XL + 6)) dbase[(size_t)kf_of(k) * 15 + lane - XL] = x;
        if constexpr (INCB) {
            const bool moved = xl_lane && pivot_real(k) && fabs(x - c_.dold) > v.wildfire;
            inc_run = __builtin_amdgcn_ballot_w64(moved) != 0 ? 0 : inc_run + 1;
        }
        xprev = x;
        STAMP(10);
    };
    {
        // (split back substitution: two waves per SIMD hide the HBM round trip between them; two panels in flight instead
        // of four keep the kernel inside the 256 registers that takes)
        PRow p3 = load_panel(n4 - 1), p2 = load_panel(n4 - 2), p1s, p0s;
        if constexpr (PD == 4) { p1s = load_panel(n4 - 3); p0s = load_panel(n4 - 4); }
        PRow& p1 = PD == 4 ? p1s : p3;
        PRow& p0 = PD == 4 ? p0s : p2;
        // increment of the keyframe after the last pivot (slot n4 & 3 = 0): zero, or the separator's / the middle system's
        double xprev = S[xl_lane ? S_DL + lane - XL : S_ZERO];
        Col ca, cb;
        if constexpr (MODE == SOLVE_FULL_BWD || INCB) {
            // two waves per SIMD: the partner wave covers this one's LDS round trips, so the recursion is not software-
            // pipelined here -- one set of column registers instead of two keeps the kernel inside 256 registers
            int stop_at = 0;
#pragma unroll 1
            for (int k = n4 - 1; k >= 3; k -= 4) {
                prep(IC<3>{}, k, p3, ca);
                solve(IC<3>{}, k, ca, xprev);
                prep(IC<2>{}, k - 1, p2, ca);
                solve(IC<2>{}, k - 1, ca, xprev);
                prep(IC<1>{}, k - 2, p1, ca);
                solve(IC<1>{}, k - 2, ca, xprev);
                prep(IC<0>{}, k - 3, p0, ca);
                solve(IC<0>{}, k - 3, ca, xprev);
                if constexpr (INCB) {
                    // keyframes k-3 .. k-1 came out as they were and every panel in front of k-3 is the old one (cg.ni: the first
                    // keyframe the forward sweep eliminated again): so would every increment in front of them
                    if (inc_run >= 3 && k - 3 <= cg.ni) { stop_at = k - 3; break; }
                }
            }
            if constexpr (INCB) { if (lane == 0) v.inc_stop[w] = lo + stop_at; }
        } else {
        prep(IC<3>{}, n4 - 1, p3, ca);
#pragma unroll 1
        for (int k = n4 - 1; k >= 3; k -= 4) {
            prep(IC<2>{}, k - 1, p2, cb);
            solve(IC<3>{}, k, ca, xprev);
            prep(IC<1>{}, k - 2, p1, ca);
            solve(IC<2>{}, k - 1, cb, xprev);
            prep(IC<0>{}, k - 3, p0, cb);
            solve(IC<1>{}, k - 2, ca, xprev);
            prep(IC<3>{}, k - 4, p3, ca);
            solve(IC<0>{}, k - 3, cb, xprev);
        }
        }
    }
#undef CBAR
#ifdef VF_SOLVE_STAMPS
    if (w == 0 && lane == 0) for (int i = (MODE == SOLVE_FULL_BWD ? 6 : 0); i < (MODE == SOLVE_FULL_BWD ? 11 : 16); i++) g_stamps[i] = st[i];
#endif
    if constexpr (MODE == SOLVE_FULL) { if (lane == 0) v.fail[w] = failed; }
    else if constexpr (MODE != SOLVE_FULL_BWD && !INCB) { if (lane == 0 && failed) atomicOr(v.fail + w, 1); }
}

__global__ void __launch_bounds__(64) k_band_solve(View v) {
    const int w = sweep_window(v, blockIdx.x);
    if (w < 0 || v.hi[w] - v.lo[w] <= 0 || window_done(v, w) || gated_off(v)) return;
    __shared__ double S[S_TOTAL];
    band_solve_body<SOLVE_FULL>(v, S, nullptr, nullptr, w, threadIdx.x, 0);
}

// the split form (see SOLVE_FULL_FWD / SOLVE_FULL_BWD)
__attribute__((amdgpu_waves_per_eu(2, 2)))
__global__ void __launch_bounds__(64) k_band_forward(View v) {
    const int w = sweep_window(v, blockIdx.x);
    if (w < 0 || v.hi[w] - v.lo[w] <= 0 || window_done(v, w) || gated_off(v)) return;
    __shared__ double S[CW_TOTAL];
    band_solve_body<SOLVE_FULL_FWD>(v, S, nullptr, nullptr, w, threadIdx.x, 0);
}
// the assembling form (see SOLVE_ASM_FWD): one wave per SIMD, 39.7 KB of LDS; followed by k_band_backward
__attribute__((amdgpu_waves_per_eu(1, 1)))
__global__ void __launch_bounds__(64) k_band_forward_asm(View v) {
    const int w = sweep_window(v, blockIdx.x);
    if (w < 0 || v.hi[w] - v.lo[w] <= 0 || window_done(v, w) || gated_off(v)) return;
    __shared__ __attribute__((aligned(16))) double S[AS_TOTAL];
    band_solve_body<SOLVE_ASM_FWD>(v, S, nullptr, nullptr, w, threadIdx.x, 0);
}
// the same as two waves per window: eliminator + assembler on one LDS image (SOLVE_ASM_A / SOLVE_ASM_B).  (Each role alone
// fits the 256 registers a wave may have at two waves per SIMD -- 225 + 24 and 205 + 40; inlined into one kernel the allocator
// takes all 256 as VGPRs and spills 17 to scratch unless it is told to keep some of the budget as AGPRs.)
__attribute__((amdgpu_waves_per_eu(2, 2)))
__global__ void __launch_bounds__(128) k_band_forward_asm2(View v, int w0) {
    const int w = sweep_window(v, w0 + blockIdx.x);
    if (w < 0 || v.hi[w] - v.lo[w] <= 0 || window_done(v, w) || gated_off(v)) return;
    __shared__ __attribute__((aligned(16))) double S[AS2_TOTAL];
    // Roles.  Every SIMD should hold ONE eliminator and one assembler: the two roles of two windows share its float64 units,
    // and two eliminators on one SIMD are two critical paths in each other's way (the same kernel on the same windows: 2.56 ms
    // balanced, 3.10-3.20 ms not).  Which SIMDs the two waves of a workgroup land on follows the dispatcher's round-robin state,
    // i.e. whatever the PREVIOUS kernel left behind: after k_decide the four workgroups of a CU come out as (0, 2), (1, 3), (2, 1),
    // (3, 0) and "wave 0 eliminates" is balanced; after the gated-off K3 launch of the hybrid solve, or in another process, it
    // is not (tools/hybrid_full_probe.py, trace_context.py; a placement kernel in front moved the step between 23.3 and 26.7 ms).
    // So the workgroups of a CU agree among themselves: each claims, in a per-CU bit mask (View::place, cleared by the launch),
    // the SIMD of its wave 0 for its eliminator, or -- that one taken -- the SIMD of its wave 1, and swaps its roles then.  The
    // waves of a CU's workgroups form cycles over its four SIMDs (two waves each), so the greedy claim ends with one eliminator
    // per SIMD whatever the order of arrival.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x < 4) S[AS_FLAGS + threadIdx.x] = threadIdx.x < 2 ? -1.0 : 0.0;
    int swapped = 0;
#if VF_ASM2_ROLES
    {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // XCC_ID [3:0]
        if (lane == 0) S[AS_FLAGS + 4 + wave] = (double)((hw >> 4) & 3);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int s0 = (int)S[AS_FLAGS + 4], s1 = (int)S[AS_FLAGS + 5];
            unsigned* cell = v.place + ((xcc & 7) << 8 | ((hw >> 8) & 0xff));               // (xcc, se, sh, cu)
            int sw = 0;
            if (s0 != s1) {
                const unsigned o0 = atomicOr(cell, 1u << s0);
                if (o0 & (1u << s0)) {
                    const unsigned o1 = atomicOr(cell, 1u << s1);
                    sw = (o1 & (1u << s1)) ? 0 : 1;
                }
            }
            S[AS_FLAGS + 6] = (double)sw;
        }
        __syncthreads();
        swapped = __builtin_amdgcn_readfirstlane((int)S[AS_FLAGS + 6]);
        if (VF_ASM2_ROLES == 2) swapped = 1;          // (test build: wave 1 eliminates in every workgroup)
    }
#else
    __syncthreads();
#endif
    const bool eliminator = (wave == 0) != (swapped != 0);
#if VF_ASM2_PRIO
    // Issue priority to the eliminator: its steps are the critical path, the assembler has 40 % slack, and the two share one
    // SIMD's float64 units (solve 3.66 -> 3.50 ms at 1 024 windows; priority to the assembler: 3.63)
    if (eliminator == (VF_ASM2_PRIO == 1)) __builtin_amdgcn_s_setprio(3);
#endif
    if (eliminator) band_solve_body<SOLVE_ASM_A>(v, S, nullptr, nullptr, w, lane, 0);
    else band_solve_body<SOLVE_ASM_B>(v, S, nullptr, nullptr, w, lane, 0);
}
__attribute__((amdgpu_waves_per_eu(2, 2)))
__global__ void __launch_bounds__(64) k_band_backward(View v) {
    const int w = sweep_window(v, blockIdx.x);
    if (w < 0 || v.hi[w] - v.lo[w] <= 0 || window_done(v, w) || gated_off(v)) return;
    // the back substitution touches nothing of the trailing window [0, S_GD + 64): its LDS starts at the write sink
    __shared__ double Sb[S_TOTAL - S_DUMP];
    band_solve_body<SOLVE_FULL_BWD>(v, Sb - S_DUMP, nullptr, nullptr, w, threadIdx.x, 0);
}

// incremental updates (View::inc_*): the suffix of the window from the checkpoint in front of the first changed keyframe
__global__ void __launch_bounds__(64) k_inc_forward(View v) {
    const int w = blockIdx.x;
    const int lo = v.lo[w], hi = v.hi[w];
    if (hi - lo <= 0) return;
    const int m = inc_start(v.inc_k[w], lo);
    if (threadIdx.x == 0) v.inc_from[w] = m < hi ? m : hi;
    if (m >= hi) { if (threadIdx.x == 0) v.fail[w] = 0; return; }      // nothing changed: the factorisation stands
    __shared__ double S[S_TOTAL];
    band_solve_body<SOLVE_INC_FWD>(v, S, nullptr, nullptr, w, threadIdx.x, 0, ChunkGeom{m - lo, 0, 0});
}
__global__ void __launch_bounds__(64) k_inc_backward(View v) {
    const int w = blockIdx.x;
    const int lo = v.lo[w], hi = v.hi[w];
    if (hi - lo <= 0) return;
    const int m = inc_start(v.inc_k[w], lo);
    if (m >= hi) { if (threadIdx.x == 0) v.inc_stop[w] = hi; return; }
    __shared__ double Sb[S_TOTAL - S_DUMP];
    band_solve_body<SOLVE_INC_BWD>(v, Sb - S_DUMP, nullptr, nullptr, w, threadIdx.x, 0, ChunkGeom{0, m - lo, 0});
}

// two waves per window (see band_solve_body); windows shorter than 32 keyframes are left to wave 0 alone
__global__ void __launch_bounds__(128) k_band_solve_tw(View v) {
    const int w = blockIdx.x;
    const int n = v.hi[w] - v.lo[w];
    if (n <= 0 || window_done(v, w)) return;
    __shared__ double S2[2 * S_TOTAL];
    __shared__ double MID[MID_TOTAL];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (n < 32) {
        if (wave == 0) band_solve_body<SOLVE_FULL>(v, S2, nullptr, nullptr, w, lane, 0);
        return;
    }
    band_solve_body<SOLVE_TWISTED>(v, S2 + wave * S_TOTAL, S2 + (1 - wave) * S_TOTAL, MID, w, lane, wave);
}

// ------------------------------------------------------------------------------------ K4p
// Partitioned solve of one window by P chunks (latency form for few windows, and the per-GPU piece of
// the time-sharded smoother).  A cut keyframe b_c sits between chunks c and c+1; the separator is
//     S_c = { b_c: 15 dof, pose of b_c + 1, pose of b_c + 2 }   (27 dof: what the profile couples across a cut),
// chunk c+1 starts at keyframe b_c + 1, whose pose rows (and those of b_c + 2) are pinned: only their
// velocity / bias dof are interior to it (vf_kernels.hpp, mask_boundary_row).
//   1. k_chunk_forward : wave 0 of every chunk eliminates its pivots with the band sweep above (couplings to
//                        its left separator dropped); leaves R_c = H[S_c,S_c] + lambda I - F^T A^-1 F and the rhs.
//   2.  (same kernel)    wave 1 follows one step behind with the spike V = L_c^-1 E (E = coupling of the chunk's
//                        dof to S_{c-1}): forward substitution through the panels (LDS ring) on the matrix cores;
//                        gives -E^T A^-1 E, -E^T A^-1 g (left separator), -F^T A^-1 E (coupling
//                        S_c x S_{c-1}) and the spike rows V for step 4.
//   3. k_sep_solve     : block-tridiagonal system of the P-1 separators (27-dof blocks), from both ends.
//   4. k_chunk_rhs     : y_k -= V_k delta(S_{c-1})  in the stored panels.
//   5. k_chunk_back    : the band back substitution of every chunk, started from delta(S_c).
// Same arithmetic as one sweep up to the elimination order (a nested-dissection ordering of the same
// Cholesky factorisation); tests/test_gpu_partitioned.py compares the increments of the two forms.
// Spike of chunk c >= 1 (second wave of k_chunk_forward).  W = the not yet substituted part of E (coupling of the
// chunk's dof to its LEFT separator, 27 columns) for the next 4 keyframes, V_k = L_kk^-1 W_k, all kept as 16x16
// tiles in the MFMA accumulator layout (row = (lane>>4) + 4r, column = lane & 15), which is also the B-operand
// layout of v_mfma_f64_16x16x4: results feed the next product without leaving registers.  Two column tiles cover
// the 27 separator columns.  The panel of step k is read from the LDS ring slot k & 3 once the sweep has published it.
__device__ __forceinline__ void chunk_spike(const View& v, double* S, int w, int c, ChunkGeom cg, int lane) {
    const int lo = v.lo[w];
    const int li = lane & 15, lq = lane >> 4;
    const size_t base = (size_t)w * v.M + lo + cg.i0;
    const double* __restrict__ Hb = v.H + (base) * HROW;
    double* __restrict__ Vb = v.Vp + base * VROW;

    // E[(kk, a)][j]: dof a of the chunk's keyframe kk (window keyframe i0 + kk) against separator column j:
    //   j < 15: dof j of the cut keyframe i0 - 1;  15..20: pose of keyframe i0 (kk = 0);  21..26: pose of i0 + 1 (kk = 1).
    // Rows: all 15 dof for kk >= 2, the velocity / bias dof (a >= 6) for the pinned head keyframes kk = 0, 1.
    // Entries come from the block rows of H (block d of row k = H[k][k-d]; d = 1 full 15x15, d = 2, 3 pose x pose),
    // the diagonal block for a keyframe against its own pose, and the transposed d = 1 block of row i0 + 1 for
    // (kk = 0 velocity/bias) x (pose of i0 + 1).  Nothing beyond kk = 4.
    auto e_val = [&](int kk, int a, int j) -> double {
        if (j >= SEP || a >= 15 || kk >= cg.ni || kk > 4) return 0.0;
        if (kk <= 1 && a < 6) return 0.0;
        const int kc = j < 15 ? -1 : (j < 21 ? 0 : 1), cc = j < 15 ? j : (j < 21 ? j - 15 : j - 21);
        const int d = kk - kc;
        if (d > 3) return 0.0;
        if (d == 0) return Hb[(size_t)kk * HROW + H_D0 + h_tri(a, cc)];              // own diagonal block: (vel/bias) x pose, a > cc
        if (d < 0) return Hb[(size_t)kc * HROW + H_D1 + cc * 15 + a];                // kk = 0, kc = 1: H[i0+1][i0] transposed
        if (d == 1) return Hb[(size_t)kk * HROW + H_D1 + a * 15 + cc];
        return (a < 6 && cc < 6) ? Hb[(size_t)kk * HROW + (d == 2 ? H_D2 : H_D3) + a * 6 + cc] : 0.0;
    };
    d4_t Wa[2], Wb[2], Wcd[2];   // W of keyframes k, k+1 and rows 0..7 of k+2 | k+3 stacked at tile rows 0..7 | 8..15
#pragma unroll
    for (int J = 0; J < 2; J++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int a = lq + 4 * r, j = 16 * J + li;
            Wa[J][r] = e_val(0, a, j);
            Wb[J][r] = e_val(1, a, j);
            Wcd[J][r] = r < 2 ? e_val(2, a, j) : e_val(3, lq + 4 * (r - 2), j);
        }
    }
    d4_t acc[3];                 // sum_k V_k^T [V_k | y_k], tiles (0,0) (0,1) (1,1)
#pragma unroll
    for (int t = 0; t < 3; t++) acc[t] = (d4_t){0, 0, 0, 0};

    // panel rows: 0..14 L[k+1][k], 15..20 L[k+2 pose][k], 21..26 L[k+3 pose][k], 27 y_k, 28..42 L_kk^-T
    // (LDS ring slot: row stride 15, cell 645 = 0.0 for the masked lanes)
    const int r23 = li < 6 ? 15 + li : ((li >= 8 && li < 14) ? 21 + li - 8 : -1);
    int o_linv[4], o_x1[4], o_x23[4], o_y[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int cq = 4 * q + lq;
        const bool ck = cq < 15;
        o_linv[q] = (ck && li < 15) ? (28 + cq) * 15 + li : 645;
        o_x1[q] = (ck && li < 15) ? li * 15 + cq : 645;
        o_x23[q] = (ck && r23 >= 0) ? r23 * 15 + cq : 645;
        o_y[q] = (ck && li == 11) ? 27 * 15 + cq : 645;          // column 27 = tile 1, lane & 15 == 11
    }
#pragma unroll 1
    for (int k = 0; k < cg.ni; k++) {
        {
            int spin = 0;
#ifdef VF_RING_WITHHOLD     // stamp / fault-injection build only: the follower of chunk 1 of window 0 never sees its producer
            if (w == 0 && c == 1) spin = RING_SPIN_MAX;
            else
#endif
            while (lds_peek(S + S_PROG) < (double)(k + 1) && lds_peek(S + S_RING_OUT) == 0.0 && ++spin < RING_SPIN_MAX) __builtin_amdgcn_s_sleep(1);
            if (spin >= RING_SPIN_MAX) {
                // the producer never published panel k: this chunk's solve is reported failed (k_decide rejects the trial, the
                // window's other chunks and every other window are untouched); from here on neither wave waits for the other
                if (lane == 0) { S[S_RING_OUT] = 1.0; atomicOr(v.fail + w, 1); }
            }
        }
        const double* Pk = S + S_P + (k & 3) * RING_SLOT;
        double linv[4], x1[4], x23[4], yv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { linv[q] = Pk[o_linv[q]]; x1[q] = -Pk[o_x1[q]]; x23[q] = -Pk[o_x23[q]]; yv[q] = Pk[o_y[q]]; }
        WSYNC();
        if (lane == 0) S[S_CONS] = (double)(k + 1);      // slot may be reused
        d4_t V[2];
#pragma unroll
        for (int J = 0; J < 2; J++) {
            V[J] = (d4_t){0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 4; q++) V[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(linv[q], Wa[J][q], V[J], 0, 0, 0);
        }
#pragma unroll
        for (int J = 0; J < 2; J++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                Wb[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[q], V[J][q], Wb[J], 0, 0, 0);
                Wcd[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(x23[q], V[J][q], Wcd[J], 0, 0, 0);
            }
        }
        // V^T [V | y]: y_k rides in column 27 (tile 1, lane & 15 == 11), where V is identically zero
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double v1y = V[1][q] + yv[q];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(V[0][q], V[0][q], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(V[0][q], v1y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(V[1][q], v1y, acc[2], 0, 0, 0);
        }
        // spike rows of keyframe k -> HBM (k_chunk_rhs reads them back)
        double* Vk = Vb + (size_t)k * VROW;
#pragma unroll
        for (int J = 0; J < 2; J++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int a = lq + 4 * r;
                if (a < 15) Vk[a * 32 + 16 * J + li] = V[J][r];
            }
        if (k + 1 < cg.ni) {
            // advance one keyframe.  E rows that only now get a register home: rows 8..14 of keyframe k+2 (it becomes
            // the "k+1" tile) and rows 0..7 of keyframe k+4 (the new "k+3" half tile); both vanish beyond keyframe 4
#pragma unroll
            for (int J = 0; J < 2; J++) {
                const int j = 16 * J + li;
                double e2 = 0.0, e3 = 0.0, f0 = 0.0, f1 = 0.0;
                if (k <= 2) { e2 = e_val(k + 2, lq + 8, j); e3 = e_val(k + 2, lq + 12, j); }
                if (k == 0) { f0 = e_val(4, lq, j); f1 = e_val(4, lq + 4, j); }
                Wa[J] = Wb[J];
                Wb[J] = (d4_t){Wcd[J][0], Wcd[J][1], e2, e3};
                Wcd[J] = (d4_t){Wcd[J][2], Wcd[J][3], f0, f1};
            }
        }
    }
    // ---- outputs --------------------------------------------------------------------------------
    double* Sm = v.sepS + ((size_t)c * v.B + w) * SEPK;  // -(V^T V), -(V^T y): added to separator c-1
    auto put = [&](const d4_t& t, int I, int J) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = 16 * I + lq + 4 * r, j = 16 * J + li;
            if (i < SEP && j < 28) {
                Sm[i * 28 + j] = -t[r];
                if (I != J && j < SEP) Sm[j * 28 + i] = -t[r];
            }
        }
    };
    put(acc[0], 0, 0); put(acc[1], 0, 1); put(acc[2], 1, 1);
    if (cg.has_sep) {
        // what is left in W belongs to the right separator: the cut keyframe (15 rows), pose rows of the two after it
        double* Cm = v.sepC + ((size_t)c * v.B + w) * SEPK;
#pragma unroll
        for (int J = 0; J < 2; J++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int a = lq + 4 * r, j = 16 * J + li;
                if (j < SEP) {
                    if (a < 15) Cm[a * SEP + j] = Wb[J][r];
                    if (r < 2 && a < 6) {
                        Cm[(15 + a) * SEP + j] = Wcd[J][r];
                        Cm[(21 + a) * SEP + j] = Wcd[J][r + 2];
                    }
                }
            }
    }
}

// wave 0: forward sweep of the chunk; wave 1 (chunks c >= 1): the spike, one step behind, fed from the LDS ring
__global__ void __launch_bounds__(128) k_chunk_forward(View v) {
    const int P = v.P, w = blockIdx.x / P, c = blockIdx.x - w * P;
    const int n = v.hi[w] - v.lo[w];
    if (n <= 0 || window_done(v, w) || gated_off(v)) return;
    const int Pe = chunk_count(n, P, v.P_fit);
    int oc0, oc1;
    own_chunks(v, Pe, oc0, oc1);
    if (c >= Pe || c < oc0 || c >= oc1) return;
    __shared__ double S[S_TOTAL_RING];
    const ChunkGeom cg = chunk_geom(n, Pe, c);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 4) S[S_P + threadIdx.x * RING_SLOT + 645] = 0.0;   // zero cell of every ring slot
    if (threadIdx.x == 4) S[S_PROG] = 0.0;
    if (threadIdx.x == 5) S[S_CONS] = 0.0;
    if (threadIdx.x == 6) S[S_RING_OUT] = 0.0;
    __syncthreads();
    if (wave == 0) band_solve_body<SOLVE_CHUNK_FWD>(v, S, nullptr, nullptr, w, lane, 0, cg, v.sepR + ((size_t)c * v.B + w) * SEPK);
    else if (c > 0) chunk_spike(v, S, w, c, cg, lane);
}
__global__ void __launch_bounds__(64) k_chunk_back(View v) {
    const int P = v.P, w = blockIdx.x / P, c = blockIdx.x - w * P;
    const int n = v.hi[w] - v.lo[w];
    if (n <= 0 || window_done(v, w) || gated_off(v)) return;
    const int Pe = chunk_count(n, P, v.P_fit);
    int oc0, oc1;
    own_chunks(v, Pe, oc0, oc1);
    if (c >= Pe || c < oc0 || c >= oc1) return;
    __shared__ double S[S_TOTAL];
    band_solve_body<SOLVE_CHUNK_BWD>(v, S, nullptr, nullptr, w, threadIdx.x, 0, chunk_geom(n, Pe, c), nullptr);
}

// y_k -= V_k delta(left separator) for every interior keyframe of the chunks c >= 1
__global__ void __launch_bounds__(256) k_chunk_rhs(View v) {
    const int P = v.P, w = blockIdx.x / P, c = blockIdx.x - w * P;
    const int lo = v.lo[w], n = v.hi[w] - lo;
    if (n <= 0 || window_done(v, w) || gated_off(v)) return;
    const int Pe = chunk_count(n, P, v.P_fit);
    int oc0, oc1;
    own_chunks(v, Pe, oc0, oc1);
    if (c == 0 || c >= Pe || c < oc0 || c >= oc1) return;
    const ChunkGeom cg = chunk_geom(n, Pe, c);
    const size_t base = (size_t)w * v.M + lo + cg.i0;
    __shared__ double dl[SEP];
    // left separator: the cut keyframe i0 - 1 (15), pose of i0, pose of i0 + 1
    if (threadIdx.x < SEP) {
        const int j = threadIdx.x;
        dl[j] = j < 15 ? v.delta[(base - 1) * 15 + j] : (j < 21 ? v.delta[base * 15 + j - 15] : v.delta[(base + 1) * 15 + j - 21]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cg.ni * 15; e += 256) {
        const int k = e / 15, a = e - k * 15;
        const double* Vr = v.Vp + (base + k) * VROW + a * 32;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < SEP; j += 3) {
            s0 = fma(Vr[j], dl[j], s0);
            s1 = fma(Vr[j + 1], dl[j + 1], s1);
            s2 = fma(Vr[j + 2], dl[j + 2], s2);
        }
        v.Lp[(base + k) * PANEL + panel_idx(27, a)] -= (s0 + s1) + s2;
    }
}

// Block-tridiagonal Cholesky of the separator chain (27-dof blocks), one workgroup per window: two teams of two
// waves eliminate the chain from both ends towards the middle separator h = m / 2 (no extra arithmetic: the
// two-sided order of the same factorisation), then back-substitute outwards from it.
// Forward, per team: step j factors the 55-row panel of its pivot separator, one row per lane of the team's
// wave 0, 27 columns in registers:
//     0..26  D (pivot block)      27..53  C (coupling to the next separator of this team's direction)      54  rhs
// (team 1 walks the chain backwards, so its coupling rows are the stored blocks transposed).
// Column operations as in the band solver: after 27 pivots the rows hold L, Z = C L^-T and y; pivots and
// multipliers are broadcast by v_readlane (the pivot rows are lanes 0..26), so a step has three LDS-only
// barriers and none inside the pivot loop.  Wave 1 of a team owns no rows: it stages the inputs of the team's
// next step (global -> registers -> LDS) while wave 0 factors.
// D_next -= Z Z^T, rhs_next -= Z y: four 16x16 MFMA tiles, two per wave, operands from the Z rows in LDS.
// The middle separator receives both teams' Schur terms and is factored by team 0.
// Backward: wave 0 of a team solves L^T delta = y - Z^T delta(neighbour towards the middle) (increment
// broadcast by v_readlane, no barriers inside a separator) while its wave 1 copies the next factor from
// HBM into LDS.  Measured cost and what bounds it: DESIGN.md "K4p".
constexpr int ZS = 28;                       // LDS row stride of the D / C panel input (27 columns + rhs)
constexpr int LXS = 64;                      // HBM: factor of one separator, column-major [27][64]: rows 0..26 L, 27..53 Z, 54 y
constexpr int FS = 57;                       // LDS row stride of the factor copy (odd: conflict-free row walks)
constexpr int ZZ = 29;                       // LDS row stride of the Z rows (odd; zero padding = MFMA K and tile remainders)
constexpr int SEP_FW = 32 * ZZ + 56 * ZS;             // forward LDS of one team: Z rows, panel input
constexpr int SEP_BW = 2 * SEP * FS;                  // backward LDS of one team: two factor copies
constexpr int SEP_LDS = 2 * (SEP_FW > SEP_BW ? SEP_FW : SEP_BW);
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for
// the stager's global loads and the factor stores at every one of the 27 pivot barriers of a step
#define LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#ifdef VF_SOLVE_STAMPS
__device__ unsigned long long g_sep_stamps[16];
#define SSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long _t; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); __builtin_amdgcn_sched_barrier(0); sst[i] += _t - stprev; stprev = _t; } while (0)
#else
#define SSTAMP(i) do {} while (0)
#endif
static_assert(SEPL >= SEP * LXS, "factor block does not fit its HBM slot");
// the three keyframes of separator `s`: element j of its 27-dof increment -> offset in v.delta relative to the cut keyframe
VF_DI int sep_delta_offset(int j) { return j < 15 ? j : (j < 21 ? 15 + (j - 15) : 30 + (j - 21)); }
__global__ void __launch_bounds__(256) k_sep_solve(View v) {
    const int P = v.P, w = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar branches on the roles
    const int team = wave >= 2 ? 1 : 0, tw = wave - 2 * team, tt = tid - 128 * team;
    const int lo = v.lo[w], n = v.hi[w] - lo;
    if (n <= 0 || window_done(v, w) || gated_off(v)) return;
    const int Pe = chunk_count(n, P, v.P_fit), m = Pe - 1;
    if (m <= 0) return;
    const int h = m / 2;                         // middle separator; team 0: 0 .. h-1 then h, team 1: m-1 .. h+1
    const int nreal = team == 0 ? h : m - 1 - h; // real forward steps of this team (the loop runs h times)
    __shared__ __attribute__((aligned(16))) double smem[SEP_LDS];
    __shared__ double dnext_s[2][32];
    double* Zs = smem + team * SEP_FW;           // [32][ZZ]: Z rows 0..26, y = row 27; rows 28..31 and columns 27, 28 stay zero
    double* Dn = Zs + 32 * ZZ;                   // [56][ZS]: panel input: D rows 0..26, C rows 27..53, rhs row 54, zero row 55
    // separator blocks are stored chunk-major, [P][B][..]: the chunks of one rank of a time-sharded window are
    // contiguous (all-gather slices); element (c, w) of this window sits c * cs (resp. c * cc) further on
    const size_t cs = (size_t)v.B * SEPK, cc = cs;
    const double* __restrict__ R = v.sepR + (size_t)w * SEPK;
    const double* __restrict__ Sx = v.sepS + (size_t)w * SEPK;          // chunk 0 is never written: 756 zeros
    const double* __restrict__ Cx = v.sepC + (size_t)w * SEPK;
    double* __restrict__ Lx = v.sepL + (size_t)w * P * SEPL;
    int failed = 0;
    // ---- chain geometry of this team: step j eliminates separator piv(j); its panel needs
    //   D  = R[piv] + S[piv+1]            (the middle's own D is carried by team 0; team 1 adds Schur terms only)
    //   C  = coupling to the next separator of the walk: sepC[piv+1] (team 0) / sepC[piv]^T (team 1)
    auto piv = [=](int j) { return team == 0 ? j : m - 1 - j; };
    auto d_ptrs = [=](int sidx, const double*& pa, const double*& pb) {      // D and rhs of separator sidx, or zeros
        const bool own = sidx >= 0 && sidx < m && (team == 0 ? sidx <= h : sidx > h);
        pa = own ? R + (size_t)sidx * cs : Sx;
        pb = (own && sidx + 1 <= m) ? Sx + (size_t)(sidx + 1) * cs : Sx;
    };
    auto c_ptr = [=](int sidx) -> const double* {   // coupling rows of the step whose pivot is sidx, or zeros
        if (team == 0) return (sidx >= 0 && sidx < h) ? Cx + (size_t)(sidx + 1) * cc : Sx;
        return (sidx > h && sidx < m) ? Cx + (size_t)sidx * cc : Sx;
    };
    for (int e = tt; e < 56 * ZS; e += 128) Dn[e] = 0.0;
    for (int e = tt; e < 32 * ZZ; e += 128) Zs[e] = 0.0;
    __syncthreads();
    // staged element (i, j) of a [27][28] (D | rhs) block / of a [27][27] coupling block -> panel input
    auto dst_d = [](int e) { const int i = e / 28, j = e - i * 28; return e < SEP * 28 ? (j == 27 ? 54 * ZS + i : i * ZS + j) : 55 * ZS + 27; };
    auto dst_c = [=](int e) {
        const int i = e / SEP, j = e - i * SEP;
        if (e >= SEP * SEP) return 55 * ZS + 27;                             // (row 55, column 27) is never read
        return team == 0 ? (27 + i) * ZS + j : (27 + j) * ZS + i;            // team 1: the stored block transposed
    };
    {
        const double *pa, *pb;
        d_ptrs(piv(0), pa, pb);
        const double* pc = c_ptr(piv(0));
        const bool any = nreal > 0 || team == 0;          // team 0 also prepares the middle when it has no step of its own
        if (any) {
            for (int e = tt; e < SEP * 28; e += 128) Dn[dst_d(e)] = pa[e] + pb[e];
            for (int e = tt; e < SEP * SEP; e += 128) Dn[dst_c(e)] = pc[e];
        }
    }
    __syncthreads();
    const int prow = (lane < 55 ? lane : 55) * ZS;     // this lane's panel row in Dn (55 = zeros)
#ifdef VF_SOLVE_STAMPS
    unsigned long long sst[16] = {0}, stprev = __builtin_amdgcn_s_memtime();
#endif
    // one elimination step of this team (real = 0: keep in step with the other team's barriers, touch nothing)
    auto step = [&](int sidx, int snext, bool real, bool more) {
        SSTAMP(0);
        if (!real) {
            LDS_BARRIER();
            LDS_BARRIER();
            LDS_BARRIER();
            return;
        }
        if (tw == 1) {
            // ---- stager: D and C of the team's next step.  Absent terms read the zero slot, so the loads are
            // unconditional and nothing is computed on them before the pivot barriers are behind us.
            int sz;
            asm volatile("v_mov_b32 %0, 0" : "=v"(sz));      // opaque zero: keeps the element maps inside the loop
            const double *pa, *pb;
            d_ptrs(more ? snext : -1, pa, pb);
            const double* pc = more ? c_ptr(snext) : Sx;
            LDS_BARRIER();   // Dn consumed: from here on the panel input is free until the Schur update
            constexpr int NSL = 12;                           // 12 * 64 = 768 >= 756 elements of (D | rhs)
            double ba[NSL], bb[NSL], bc[NSL];
#pragma unroll
            for (int q = 0; q < NSL; q++) {
                const int e = lane + 64 * q + sz, ed = e < SEP * 28 ? e : 0;
                ba[q] = pa[ed];
                bb[q] = pb[ed];
                bc[q] = pc[e < SEP * SEP ? e : 0];
            }
#pragma unroll
            for (int q = 0; q < NSL; q++) {
                const int e = lane + 64 * q + sz;
                Dn[dst_d(e)] = ba[q] + bb[q];
                Dn[dst_c(e)] = bc[q];
            }
        } else {
            // ---- panel rows ----
            double p[SEP];
#pragma unroll
            for (int c = 0; c < SEP; c++) p[c] = Dn[prow + c];
            LDS_BARRIER();   // Dn consumed
            SSTAMP(1);
            // the 27 pivot rows are lanes 0..26 of this wave: pivots and multipliers by v_readlane, no LDS round trip and no
            // barrier inside the step; straight-line code in a fixed issue order (tools/gen_pivot.py)
            double pv_inv;
#include "vf_pivot_27.inc"
            if (!(pv_inv < 1e300)) failed = 1;
            SSTAMP(2);
            // factor rows -> HBM, column-major: one contiguous 8-byte-per-lane store per column
            if (lane < 55) {
                double* dst = Lx + (size_t)sidx * SEPL + lane;
#pragma unroll
                for (int c = 0; c < SEP; c++) dst[c * LXS] = p[c];
            }
            if (lane >= 27 && lane <= 54) {
#pragma unroll
                for (int c = 0; c < SEP; c++) Zs[(lane - 27) * ZZ + c] = p[c];
            }
        }
        LDS_BARRIER();
        SSTAMP(3);
        if (more) {
            // D_next -= Z Z^T, rhs_next -= Z y on the matrix cores: 2 x 2 tiles of 16 x 16, two per wave,
            // K = 27 (7 steps of 4); both operands come from the Z rows in LDS (row 27 = y)
            const int li = lane & 15, lq = lane >> 4;
#pragma unroll 1
            for (int t = tw; t < 4; t += 2) {
                const int I = t >> 1, J = t & 1;
                // rows 27 (y), 28..31 of tile I = 1 produce output rows that are dropped; zero padding does the masking
                const double* za = Zs + (16 * I + li) * ZZ + lq;
                const double* zb = Zs + (16 * J + li) * ZZ + lq;
                double av[7], bv[7];
#pragma unroll
                for (int q = 0; q < 7; q++) { av[q] = za[4 * q]; bv[q] = zb[4 * q]; }
                d4_t acc = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 7; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = 16 * I + lq + 4 * r, j = 16 * J + li;
                    if (i < 27 && j < 28) Dn[j == 27 ? 54 * ZS + i : i * ZS + j] -= acc[r];
                }
            }
        }
        LDS_BARRIER();
    };
#pragma unroll 1
    for (int j = 0; j < h; j++) {
        const bool real = j < nreal;
        const int sidx = piv(j), snext = team == 0 ? sidx + 1 : sidx - 1;
        step(sidx, snext, real, true);
    }
    // ---- the middle separator: team 0's panel input (its own D + left Schur terms) + team 1's Schur terms
    {
        const double* Dother = smem + SEP_FW + 32 * ZZ;
        if (team == 0)
            for (int e = tt; e < 55 * ZS; e += 128) Dn[e] += (e < 27 * ZS || e >= 54 * ZS) ? Dother[e] : 0.0;
        LDS_BARRIER();
        step(h, -1, team == 0, false);
    }
    SSTAMP(4);
    __syncthreads();   // the factor blocks in HBM (written by other threads of the workgroup) are read back below
    // ---- back substitution outwards from the middle ------------------------------------------------
    // HBM -> LDS copy of one factor block: [27 columns][55 rows], rows contiguous -> F[c * FS + r]
    // (all loads of a thread are issued before the first LDS write: 27 independent round trips, not 27 serial ones)
    double* Fb = smem + team * SEP_BW;
    auto copy_factor = [&](int sidx, int buf) {     // the 64 lanes of a team's wave 1
        const double* Ls = Lx + (size_t)sidx * SEPL;
        double x[SEP];
#pragma unroll
        for (int q = 0; q < SEP; q++) x[q] = Ls[q * LXS + lane];
#pragma unroll
        for (int q = 0; q < SEP; q++)
            if (lane < 55) Fb[buf * SEP * FS + q * FS + lane] = x[q];
    };
    // iteration i: team 0 solves separator h - i (i = 0: the middle, nothing towards the middle to subtract),
    // team 1 solves h + i (from i = 1); factor copies live in buffer i & 1
    auto mine = [=](int i) { return team == 0 ? h - i : h + i; };
    auto have = [=](int i) { return team == 0 ? i <= h : (i >= 1 && h + i <= m - 1); };
    if (tw != 0) {
        if (team == 0) copy_factor(h, 0);
        else if (have(1)) copy_factor(h + 1, 1);
    }
    if (tt < 32) dnext_s[team][tt] = 0.0;
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i <= h; i++) {
        if (tw != 0) {
            if (have(i + 1) && !(team == 1 && i == 0)) copy_factor(mine(i + 1), (i + 1) & 1);
        } else if (have(i)) {
            // lane j < 27 owns column j: F[j * FS + r] = L[r][j] (r >= j), Z[q][j] at row 27 + q, y_j at row 54
            const double* F = Fb + (i & 1) * SEP * FS + (lane < 27 ? lane : 0) * FS;
            double t0 = F[54], t1 = 0.0, t2 = 0.0;
            if (i > 0) {
#pragma unroll
                for (int q = 0; q < SEP; q += 3) {
                    t0 = fma(-F[27 + q], dnext_s[team][q], t0);
                    t1 = fma(-F[27 + q + 1], dnext_s[team][q + 1], t1);
                    t2 = fma(-F[27 + q + 2], dnext_s[team][q + 2], t2);
                }
            }
            double t = (t0 + t1) + t2;
            double Lc[SEP];
#pragma unroll
            for (int c = 0; c < SEP; c++) Lc[c] = F[c];
            const double invd = 1.0 / F[lane < 27 ? lane : 0];       // 1 / L_jj
            // L^T x = t, last unknown first: x_c = t_c / L_cc, then t_j -= L[c][j] x_c for j < c
#pragma unroll
            for (int c = SEP - 1; c >= 0; c--) {
                const double xc = readlane_d(t * invd, c);
                t = lane == c ? xc : (lane < c ? fma(-Lc[c], xc, t) : t);
            }
            if (lane < 27) {
                dnext_s[team][lane] = t;
                if (i == 0) dnext_s[1][lane] = t;      // the middle's increment starts team 1's walk as well
                const ChunkGeom cg = chunk_geom(n, Pe, mine(i));
                v.delta[((size_t)w * v.M + lo + cg.i0 + cg.ni) * 15 + sep_delta_offset(lane)] = t;
            }
        }
        __syncthreads();
    }
    SSTAMP(5);
#ifdef VF_SOLVE_STAMPS
    if (w == 0 && tid == 0) for (int i = 0; i < 16; i++) g_sep_stamps[i] = sst[i];
#endif
    if (failed && tt == 0) atomicOr(v.fail + w, 1);
}

#ifdef VF_SOLVE_STAMPS
extern "C" int vf_debug_sep_stamps(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sep_stamps), 16 * sizeof(unsigned long long));
}
extern "C" int vf_debug_k3_stamps(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k3_stamps), 8 * sizeof(unsigned long long));
}
extern "C" int vf_debug_k3_loop(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k3_loop), 8 * sizeof(unsigned long long));
}
extern "C" int vf_debug_solve_stamps(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 16 * sizeof(unsigned long long));
}
#endif

// ------------------------------------------------------------------------------------ K5
__global__ void __launch_bounds__(256) k_retract(View v) {
    const long gk = (long)blockIdx.x * 256 + threadIdx.x;
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    if (k < v.lo[w] || k >= v.hi[w] || window_done(v, w)) return;
    const int b = v.sel[w];
    const State s = load_state(v, b, gk);
    const double* d = v.delta + (size_t)gk * 15;
    State o;
    Q4 dq;
    V3 dtv;
    se3_exp(v3(d[0], d[1], d[2]), v3(d[3], d[4], d[5]), &dq, &dtv);
    o.q = qnormalize(qmul(s.q, dq));
    o.t = s.t + mul(qrot(s.q), dtv);
    o.vel = s.vel + v3(d[6], d[7], d[8]);
    o.ba = s.ba + v3(d[9], d[10], d[11]);
    o.bg = s.bg + v3(d[12], d[13], d[14]);
    store_state(v, b ^ 1, gk, o);
}

// cost of buffer (sel ^ !init) per window, then the LM decision.  Block = window, 256 threads (1024 when there are few
// windows: the per-thread loop over a long window's factors is a latency chain); fixed-shape tree reduction, so the
// cost is bitwise reproducible for a given engine.
// Time-sharded windows: every rank holds the residual of every factor (k_linearize_all<true>), so every rank forms the
// whole cost itself and takes the same decision -- no exchange; the solve-failure flags of the other ranks arrive in the
// tail of the increment buffer, summed by the same all-reduce as the increments (k_mask_delta).
__global__ void __launch_bounds__(1024) k_decide(View v, int init) {
    const int w = blockIdx.x, tid = threadIdx.x;
    if (window_done(v, w) && !init) return;
    const int lo = v.lo[w], hi = v.hi[w];
    const int b = init ? v.sel[w] : (v.sel[w] ^ 1);
    const size_t tiles = (size_t)(v.G >> 6);
    const double* imu_r = v.imu_r + (size_t)b * tiles * IMU_R * TILE;
    const double* btw_out = v.btw_out + (size_t)b * tiles * BTW_OUT * TILE;
    double s = 0.0;
    const int nt = (int)blockDim.x;
    for (int k = lo + 1 + tid; k < hi; k += nt) {
        const long gk = (long)w * v.M + k;
        const double* f = imu_r + (size_t)(gk >> 6) * IMU_R * TILE + (gk & 63);
        double c = 0.0;
#pragma unroll
        for (int r = 0; r < 15; r++) { const double x = f[(size_t)r * TILE]; c = fma(x, x, c); }
        const int a = v.btw_a[gk];
        if (a >= lo && a < k) {
            const double* fb = btw_out + (size_t)(gk >> 6) * BTW_OUT * TILE + (gk & 63);
#pragma unroll
            for (int r = 0; r < 6; r++) { const double x = fb[(size_t)r * TILE]; c = fma(x, x, c); }
        }
        s += c;
    }
    if (tid == 0) {
        const int pk = v.prior_k[w];
        if (pk >= lo && pk < hi) {
            const double* f = v.prior_out + ((size_t)b * v.B + w) * PRIOR_OUT;
            for (int r = 0; r < 15; r++) s = fma(f[r], f[r], s);
        }
        if (v.mp_on[w] && hi - lo >= 3) s += 2.0 * v.mp_out[((size_t)b * v.B + w) * 28 + 27];
        for (int xs = 0; xs < v.x_max; xs++) {           // far between factors, nonlinear and linear (empty slots hold zeros)
            const double* f = v.x_out + (((size_t)b * v.B + w) * v.x_max + xs) * BTW_OUT;
            const double* fl = v.xl_out + ((size_t)b * v.B + w) * 6 * v.x_max + 6 * xs;
            for (int r = 0; r < 6; r++) s = fma(f[r], f[r], fma(fl[r], fl[r], s));
        }
    }
    __shared__ double red[1024];
    __shared__ int outcome_s;
    red[tid] = s;
    __syncthreads();
    for (int st = nt >> 1; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    // outcome of the trial: 0 rejected, 1 accepted, 2 kept provisionally (non-monotone LM: the cost rose, the excursion goes
    // on), 3 the excursion failed (the point it started from is restored)
    if (tid == 0) {
        const double c = 0.5 * red[0];
        if (v.sh_G > 1 && !init) v.fail[w] = v.delta[(size_t)v.G * 15 + w] > 0.0 ? 1 : 0;   // a failed elimination on any rank rejects the trial
        if (init) {
            v.cost[w] = c;
            v.fail[w] = 0;
            if (!(v.fresh[w] >= 2 && v.fresh[w] < 64)) v.fresh[w] = 1;   // (a warm start has marked the window "ends only")
            if (v.stop_on) v.done[w] = 0;
            if (v.nm_W > 0) v.prov[w] = 0;
            outcome_s = -1;
        } else {
            // accept unless the cost rises by more than the rounding floor of its own evaluation (View::accept_rel): at a
            // converged window a strict "c < cost" is decided by the last bits of two 1000-term sums, and a rejected
            // Newton step leaves the soft modes of the window where they were (DESIGN.md "Accept rule at the floor")
            const int prov = v.nm_W > 0 ? v.prov[w] : 0;
            const double refc = prov > 0 ? v.ref_cost[w] : v.cost[w];        // an excursion is judged against the point it left
            const bool solved = v.fail[w] == 0;
            const bool ok = solved && (c < refc + v.accept_rel * refc);
            const int outcome = ok ? 1 : (v.nm_W > 0 && solved && prov < v.nm_W) ? 2 : (prov > 0 ? 3 : 0);
            outcome_s = outcome | (prov << 8);
            if (!solved) v.n_fail[w] += 1;
            v.fresh[w] = outcome ? 1 : 0;
            if (v.stop_on && solved && outcome < 2) {
                // gtsam checkConvergence (absolute / relative decrease of the accepted step), applied to rejected
                // trials too: a trial that changes the cost by less than the tolerance in either direction means
                // the window sits at its rounding floor, where accept / reject is decided by the last bit
                const double dec = fabs(refc - c);
                if (dec <= v.abs_tol || dec <= v.rel_tol * refc) v.done[w] = 1;
            }
            double l = v.lambda[w];
            if (outcome == 1 || outcome == 2) {
                if (outcome == 2 && prov == 0) v.ref_cost[w] = v.cost[w];
                v.sel[w] ^= 1;
                v.cost[w] = c;
                if (outcome == 1) v.n_acc[w] += 1; else v.n_prov[w] += 1;
                // (a provisional trial divides twice: an excursion is the damped iteration's way towards the Gauss-Newton
                // step, whose cost falls only once the stiff residuals its first-order move disturbed have been corrected)
                l = outcome == 1 ? l / v.lam_down : l / (v.lam_down * v.lam_down);
                v.lambda[w] = l < v.lam_min ? v.lam_min : l;
                if (v.nm_W > 0) v.prov[w] = outcome == 1 ? 0 : prov + 1;
            } else {
                v.n_rej[w] += 1;
                l *= v.lam_up;
                v.lambda[w] = l > v.lam_max ? v.lam_max : l;
                if (outcome == 3) { v.cost[w] = refc; v.prov[w] = 0; v.relin[w] = 1; }
            }
            v.fail[w] = 0;
        }
    }
    if (v.nm_W > 0 && !init) {
        __syncthreads();
        const int outcome = outcome_s & 0xff, prov = outcome_s >> 8;
        // (v.sel[w] was read into b = sel ^ 1 at the top: b ^ 1 is the buffer that was current when the trial was made)
        if (outcome == 2 && prov == 0)
            for (int e = tid; e < (hi - lo) * 16; e += nt) {
                const int cc = e / (hi - lo), k = lo + e - cc * (hi - lo);
                v.x_best[(size_t)cc * v.G + (size_t)w * v.M + k] = XS(b ^ 1, cc, (long)w * v.M + k);
            }
        if (outcome == 3)
            for (int e = tid; e < (hi - lo) * 16; e += nt) {
                const int cc = e / (hi - lo), k = lo + e - cc * (hi - lo);
                XS(b ^ 1, cc, (long)w * v.M + k) = v.x_best[(size_t)cc * v.G + (size_t)w * v.M + k];
            }
    }
}

// non-monotone LM, end of a solve: an excursion still open when the trials run out is undone (the point it started from was
// the best one seen)
__global__ void __launch_bounds__(1024) k_close_excursion(View v) {
    const int w = blockIdx.x, tid = threadIdx.x, nt = (int)blockDim.x;
    const int prov = v.prov[w], lo = v.lo[w], hi = v.hi[w], b = v.sel[w];
    __syncthreads();
    if (prov <= 0 || hi - lo <= 0) return;
    for (int e = tid; e < (hi - lo) * 16; e += nt) {
        const int cc = e / (hi - lo), k = lo + e - cc * (hi - lo);
        XS(b, cc, (long)w * v.M + k) = v.x_best[(size_t)cc * v.G + (size_t)w * v.M + k];
    }
    // (relin doubles as "this solve ended inside an excursion": the next solve then keeps the damping the excursion had reached
    // instead of starting from lambda0 again -- k_reset_lambda -- or a window that needs more trials than one solve has would
    // repeat the same first trials for ever)
    if (tid == 0) { v.cost[w] = v.ref_cost[w]; v.prov[w] = 0; v.relin[w] = 1; v.fresh[w] = 1; v.carry[w] = 1; }
}
// every solve starts from lambda0, as a fresh LevenbergMarquardtOptimizer would -- except a window whose previous solve was cut
// off inside an excursion (non-monotone LM), which goes on from the damping it had reached
__global__ void __launch_bounds__(256) k_reset_lambda(View v, const double* __restrict__ lambda0) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= v.B) return;
    if (v.carry && v.carry[w]) { v.carry[w] = 0; return; }
    v.lambda[w] = lambda0[w];
}

// ------------------------------------------------------------------------------------ a2
VF_DI State predict_state(const View& v, const State& si, long gk_factor) {
    const double* in = v.imu_in + (size_t)(gk_factor >> 6) * IMU_IN * TILE + (gk_factor & 63);
#define IN(f) in[(size_t)(f) * TILE]
    const double dt = IN(0);
    const V3 dba = si.ba - v3(IN(10), IN(11), IN(12));
    const V3 dbg = si.bg - v3(IN(13), IN(14), IN(15));
    double xt[9];
    for (int r = 0; r < 9; r++) {
        double s = IN(1 + r);
        s = fma(IN(16 + r * 6 + 0), dba.x, s);
        s = fma(IN(16 + r * 6 + 1), dba.y, s);
        s = fma(IN(16 + r * 6 + 2), dba.z, s);
        s = fma(IN(16 + r * 6 + 3), dbg.x, s);
        s = fma(IN(16 + r * 6 + 4), dbg.y, s);
        s = fma(IN(16 + r * 6 + 5), dbg.z, s);
        xt[r] = s;
    }
#undef IN
    const M3 Ri = qrot(si.q);
    const V3 grav = v3(v.grav[0], v.grav[1], v.grav[2]);
    const V3 gib = mulT(Ri, grav), vib = mulT(Ri, si.vel);
    const V3 xp = v3(xt[3], xt[4], xt[5]) + dt * vib + (0.5 * dt * dt) * gib;
    const V3 xv = v3(xt[6], xt[7], xt[8]) + dt * gib;
    State o;
    o.q = qnormalize(qmul(si.q, qexp(v3(xt[0], xt[1], xt[2]))));
    o.t = si.t + mul(Ri, xp);
    o.vel = si.vel + mul(Ri, xv);
    o.ba = si.ba;
    o.bg = si.bg;
    return o;
}

// window < 0: all windows (one lane each)
// from_trial: the first prediction starts from the TRIAL buffer's state of keyframe k0 - 1 (reference-compat solves: the
// trial buffer holds the estimate theta (+) delta the reference predicts from, GraphManager.cpp:152-153)
__global__ void k_predict(View v, int window, int k0, int n, int from_trial) {
    const int w = window >= 0 ? window : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= v.B || (window >= 0 && (blockIdx.x | threadIdx.x))) return;
    const int b = v.sel[w];
    for (int k = k0; k < k0 + n; k++) {
        const long gk = (long)w * v.M + k;
        const State si = load_state(v, (from_trial && k == k0) ? b ^ 1 : b, gk - 1);
        const State sn = predict_state(v, si, gk);
        store_state(v, b, gk, sn);
        if (from_trial) store_state(v, b ^ 1, gk, sn);      // the estimate of a keyframe that has no increment yet
    }
}

// Reference-compat solves (one iSAM2-like update per vf_solve, GraphManager.cpp:38-43,126-127): fluid relinearisation --
// a keyframe whose pending increment reaches the threshold in any component (ISAM2Params::relinearizeThreshold, a scalar:
// max |delta| >= threshold) moves its linearisation point there, theta <- theta (+) delta, delta <- 0; the others keep
// theirs.  Lane = keyframe.
__global__ void __launch_bounds__(256) k_relinearize(View v, double threshold) {
    const long gk = (long)blockIdx.x * 256 + threadIdx.x;
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    if (k < v.lo[w] || k >= v.hi[w]) return;
    double* d = v.delta + (size_t)gk * 15;
    double m = 0.0;
#pragma unroll
    for (int a = 0; a < 15; a++) m = fmax(m, fabs(d[a]));
    if (!(m >= threshold)) return;
    const int b = v.sel[w];
    const State s = load_state(v, b, gk);
    State o;
    Q4 dq;
    V3 dtv;
    se3_exp(v3(d[0], d[1], d[2]), v3(d[3], d[4], d[5]), &dq, &dtv);
    o.q = qnormalize(qmul(s.q, dq));
    o.t = s.t + mul(qrot(s.q), dtv);
    o.vel = s.vel + v3(d[6], d[7], d[8]);
    o.ba = s.ba + v3(d[9], d[10], d[11]);
    o.bg = s.bg + v3(d[12], d[13], d[14]);
    store_state(v, b, gk, o);
#pragma unroll
    for (int a = 0; a < 15; a++) d[a] = 0.0;
}

// The same for incremental updates (View::inc_*), which also need to know WHERE the first change is: the first slot whose
// linearisation point moves here, or the first of the `appended` keyframes at the window's end (their factors are new), or
// the window's first keyframe when the caller says the factorisation is void.  Only keyframes from inc_stop on have an
// increment that the last back substitution touched.  inc_k is INT_MAX when this runs (k_inc_retract leaves it so).
__global__ void __launch_bounds__(256) k_inc_begin(View v, double threshold, int appended, int invalid) {
    const int w = blockIdx.y;
    const int lo = v.lo[w], hi = v.hi[w];
    const int from = invalid ? lo : v.inc_stop[w];
    const int k = (from & ~255) + (int)(blockIdx.x * 256 + threadIdx.x);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMin(v.inc_k + w, invalid ? lo : (hi - appended > lo ? hi - appended : lo));
    if (k < lo || k < from || k >= hi) return;
    const long gk = (long)w * v.M + k;
    double* d = v.delta + (size_t)gk * 15;
    double m = 0.0;
#pragma unroll
    for (int a = 0; a < 15; a++) m = fmax(m, fabs(d[a]));
    if (!(m >= threshold)) return;
    const int b = v.sel[w];
    const State s = load_state(v, b, gk);
    State o;
    Q4 dq;
    V3 dtv;
    se3_exp(v3(d[0], d[1], d[2]), v3(d[3], d[4], d[5]), &dq, &dtv);
    o.q = qnormalize(qmul(s.q, dq));
    o.t = s.t + mul(qrot(s.q), dtv);
    o.vel = s.vel + v3(d[6], d[7], d[8]);
    o.ba = s.ba + v3(d[9], d[10], d[11]);
    o.bg = s.bg + v3(d[12], d[13], d[14]);
    store_state(v, b, gk, o);
#pragma unroll
    for (int a = 0; a < 15; a++) d[a] = 0.0;
    atomicMin(v.inc_k + w, k);
}
// the estimate theta (+) delta of the keyframes whose increment the back substitution touched, into the trial buffer (k_retract
// for the rest of the library); the last thing an incremental update does: inc_k is ready for the next one
__global__ void __launch_bounds__(256) k_inc_retract(View v) {
    const int w = blockIdx.y;
    const int lo = v.lo[w], hi = v.hi[w], from = v.inc_stop[w];
    const int k = (from & ~255) + (int)(blockIdx.x * 256 + threadIdx.x);
    if (blockIdx.x == 0 && threadIdx.x == 0) v.inc_k[w] = 0x7fffffff;
    if (k < lo || k < from || k >= hi) return;
    const long gk = (long)w * v.M + k;
    const int b = v.sel[w];
    const State s = load_state(v, b, gk);
    const double* d = v.delta + (size_t)gk * 15;
    State o;
    Q4 dq;
    V3 dtv;
    se3_exp(v3(d[0], d[1], d[2]), v3(d[3], d[4], d[5]), &dq, &dtv);
    o.q = qnormalize(qmul(s.q, dq));
    o.t = s.t + mul(qrot(s.q), dtv);
    o.vel = s.vel + v3(d[6], d[7], d[8]);
    o.ba = s.ba + v3(d[9], d[10], d[11]);
    o.bg = s.bg + v3(d[12], d[13], d[14]);
    store_state(v, b ^ 1, gk, o);
}

// Fixed-lag marginalisation of the oldest keyframe m = lo: the Schur complement of
// every factor touching m -- its prior or marginal prior, the IMU factor m -> m+1, the between
// factors starting at m -- taken at the current linearisation (buffer `sel`), onto
// [m+1: 15][m+2: pose 6][m+3: pose 6].  One 256-thread workgroup per window, 42x42 system in LDS, Gaussian
// elimination of the 15 leading columns (no square roots).  Runs once per slide.
// stash != null (FAR = false only): the new marginal prior goes into stash[w][MARG_STASH] instead of the window's own arrays, and
// nothing of the window changes -- k_marg_commit puts it in place later (vf_engine_marginalize_ahead: the marginalisation a full
// fixed-lag window will need at its next update, computed behind the solve that has just finished instead of in front of the
// next one).
constexpr int MARG_STASH = MARG_STASH_DOUBLES;
template <bool FAR>
__global__ void __launch_bounds__(256) k_marginalize(View v, int* status, double* __restrict__ stash) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const int lo = v.lo[w], hi = v.hi[w];
    if (hi - lo < 4) { if (lane == 0) atomicOr(status, 2); return; }
    const int b = v.sel[w];
    const size_t tiles = (size_t)(v.G >> 6);
    const long g0 = (long)w * v.M + lo;
    __shared__ double A[42 * 43];
    __shared__ double bv[42];
    const double* imu = v.imu_r + (size_t)b * tiles * IMU_R * TILE + (size_t)((g0 + 1) >> 6) * IMU_R * TILE + ((g0 + 1) & 63);   // residual
    const double* jbuf = v.imu_j + (size_t)b * (size_t)(v.G >> JT_LOG) * JT_STRIDE;
    const bool has_prior = v.prior_k[w] == lo;
    const bool has_mp = v.mp_on[w] != 0;
    const double* Pq = v.prior_out + ((size_t)b * v.B + w) * PRIOR_OUT;
    const double* ML = v.mp_L + (size_t)w * 729;
    const double* Mg = v.mp_out + ((size_t)b * v.B + w) * 28;
    int bd[3];   // between factor m -> m+d present?
    for (int d = 1; d <= 3; d++) bd[d - 1] = (v.btw_a[g0 + d] == lo);
    // The linearisations of the factors that touch m are staged into LDS once (the whitened Jacobian of the IMU factor
    // m -> m+1 as a dense 15 x 30 block + its residual, the three between linearisations that may start at m); every
    // entry of the 42 x 42 system then costs LDS reads.  (Read where they lie -- J-stream entries through the index
    // table, AoSoA fields -- the 54 000 dependent global loads of this phase were most of the kernel's 95 us, i.e. an
    // eighth of a GraphManager solve.)
    __shared__ double LJ[15 * 30 + 15];
    __shared__ double LBt[3 * BTW_OUT];
    // far between factors that start at m and end within the prior's reach (m+1 .. m+3) are absorbed into the marginal prior
    // like band factors: a far factor is transported from keyframe to keyframe while its anchor leaves the window (vf_engine.hip
    // "transport_far") until it is this short, and its information then outlives both its ends
    __shared__ double LXt[MAX_EXTRA * BTW_OUT];
    __shared__ int xd[MAX_EXTRA];
    if (lane < MAX_EXTRA) {
        int d = 0;
        if (lane < v.x_max) {
            const int i = w * v.x_max + lane, a = v.x_a[i], kb = v.x_b[i];
            if (a == lo && kb - lo >= 1 && kb - lo <= 3 && kb < hi) d = kb - lo;
        }
        xd[lane] = d;
    }
    __syncthreads();
    for (int e = lane; e < MAX_EXTRA * BTW_OUT; e += 256) {
        const int s = e / BTW_OUT, f = e - s * BTW_OUT;
        LXt[e] = (s < v.x_max && xd[s]) ? v.x_out[(((size_t)b * v.B + w) * v.x_max + s) * BTW_OUT + f] : 0.0;
    }
    for (int e = lane; e < 15 * 30 + 15; e += 256)
        LJ[e] = e < 450 ? jstream_entry(jbuf, g0 + 1, e / 30, e % 30) : imu[(size_t)(e - 450) * TILE];
    for (int e = lane; e < 3 * BTW_OUT; e += 256) {
        const int d = e / BTW_OUT + 1, f = e - (d - 1) * BTW_OUT;
        const long gs = g0 + d;
        LBt[e] = bd[d - 1] ? v.btw_out[(size_t)b * tiles * BTW_OUT * TILE + ((size_t)(gs >> 6) * BTW_OUT + f) * TILE + (gs & 63)] : 0.0;
    }
    __syncthreads();
    auto JI = [&](int r, int c30) { return LJ[r * 30 + c30]; };
    // index maps of the 42-vector: [m:15][m+1:15][m+2 pose][m+3 pose]
    auto imu_c = [](int i) { return i < 15 ? imu_col(0, i) : imu_col(1, i - 15); };   // i < 30
    auto mp_i = [](int i) { return i < 15 ? i : (i < 21 ? i : (i >= 30 && i < 36 ? i - 9 : -1)); };   // 42-index -> 27-index
    for (int e = lane; e < 42 * 42 + 42; e += 256) {
        const bool is_b = e >= 42 * 42;
        const int i = is_b ? e - 42 * 42 : e / 42, j = is_b ? -1 : e - (e / 42) * 42;
        double sum = 0.0;
        if (i < 30 && (is_b || j < 30)) {   // IMU factor m -> m+1
            const int ci = imu_c(i);
            for (int r = 0; r < 15; r++) sum = fma(JI(r, ci), is_b ? LJ[450 + r] : JI(r, imu_c(j)), sum);
        }
        for (int q = 0; q < 3 + MAX_EXTRA; q++) {      // between factors m -> m+d: Ja on m pose, Jb on (m+d) pose (band slots, then far slots)
            const int d = q < 3 ? (bd[q] ? q + 1 : 0) : xd[q - 3];
            if (!d) continue;
            const double* F = q < 3 ? LBt + q * BTW_OUT : LXt + (q - 3) * BTW_OUT;
            const int ob = d == 1 ? 15 : (d == 2 ? 30 : 36);
            const int ia = i < 6 ? i : -1, ib = (i >= ob && i < ob + 6) ? i - ob : -1;
            const int ja = (!is_b && j < 6) ? j : -1, jb = (!is_b && j >= ob && j < ob + 6) ? j - ob : -1;
            if (ia < 0 && ib < 0) continue;
            if (!is_b && ja < 0 && jb < 0) continue;
            for (int r = 0; r < 6; r++) {
                const double xi = ia >= 0 ? F[6 + r * 6 + ia] : F[42 + r * 6 + ib];
                const double xj = is_b ? F[r] : (ja >= 0 ? F[6 + r * 6 + ja] : F[42 + r * 6 + jb]);
                sum = fma(xi, xj, sum);
            }
        }
        if (has_prior && i < 15 && (is_b || j < 15))
            for (int r = 0; r < 15; r++) sum = fma(Pq[15 + r * 15 + i], is_b ? Pq[r] : Pq[15 + r * 15 + j], sum);
        if (has_mp) {
            const int mi = mp_i(i), mj = is_b ? 0 : mp_i(j);
            if (mi >= 0 && mj >= 0) sum += is_b ? Mg[mi] : ML[mi * 27 + mj];
        }
        if (is_b) bv[i] = sum; else A[i * 43 + j] = sum;
    }
    __syncthreads();
    // ---- FAR: the window's linear far factor (View::xl_*; every row of it has the leaving keyframe in its support) and the
    // nonlinear far factors anchored at m whose end lies beyond the prior's reach are marginalised WITH m, jointly and exactly
    // at the current linearisation.  With W = [W_mn | W_b | r] their whitened rows over (m and the prior's keyframes n: the 42
    // columns of A; the far ends b: 6 each; residual), the system to eliminate m from is
    //     [A + W_mn^T W_mn   (W_b^T W_mn)^T]         [bv + W_mn^T r]
    //     [W_b^T W_mn         W_b^T W_b    ]    and  [W_b^T r      ]
    // whose far-end rows are kept in E (the A block and bv in place).  A far end that has come within the prior's reach
    // (m + 3) is folded into the columns of that keyframe first.  After the 15 pivots, (n, b) hold the exact marginal S; it is
    // split again into a prior on n alone and six rows per far end:
    //     S_bb = L L^T,  X = L^-1 [S_bn | eta_b],  rows U' = [X_n | L^T], r' = x_eta;  prior: S_nn - X_n^T X_n, eta_n - X_n^T x_eta
    // (U'^T U' + prior = S).  Rows and far ends are written back compacted; the far ends' linearisation points move to the
    // current states, like the prior's.
    constexpr int FR = FAR ? 6 * MAX_EXTRA : 1, FC = 42 + 6 * MAX_EXTRA + 1;      // rows of W / E, their columns
    __shared__ double Wf[FR * FC], E[FR * FC];
    __shared__ int f_kb[MAX_EXTRA], f_src[MAX_EXTRA], f_fold[MAX_EXTRA], f_T, f_live;
    if constexpr (FAR) {
        const int nl_old = v.xl_n[w];
        if (lane == 0) {
            // far ends of the joint factor: the old linear ones, then the nonlinear far factors this marginalisation converts
            // (f_src: -1 - e = old far end e, i >= 0 = entry i of the nonlinear list)
            int T = 0;
            for (int e = 0; e < nl_old; e++) { f_kb[T] = v.xl_b[w * v.x_max + e]; f_src[T] = -1 - e; f_fold[T] = f_kb[T] == lo + 3; T++; }
            for (int i = 0; i < v.x_max && T < MAX_EXTRA; i++) {
                const int a = v.x_a[w * v.x_max + i], kb = v.x_b[w * v.x_max + i];
                if (a == lo && kb - lo > 3 && kb < hi) { f_kb[T] = kb; f_src[T] = i; f_fold[T] = 0; T++; }
            }
            f_T = T;
        }
        __syncthreads();
        const int T = f_T, R = 6 * T, NC = 42 + 6 * T + 1;       // (columns: 0..41 as A, 42 + 6 t + c far end t, NC - 1 residual)
        for (int e = lane; e < R * FC; e += 256) {
            const int row = e / FC, c = e - row * FC, t = row / 6, j = row - 6 * t;
            double x = 0.0;
            if (c < NC) {
                const bool res = c == NC - 1;
                const int tc = (c >= 42 && !res) ? (c - 42) / 6 : -1, cc = tc >= 0 ? c - 42 - 6 * tc : 0;
                if (f_src[t] < 0) {
                    // row (6 e + j) of the old factor: columns [m: 0..14][m+1 pose: 15..20][m+2 pose: 21..26][far end q: 27 + 6 q ..]
                    const int eo = -1 - f_src[t];
                    const double* U = v.xl_U + ((size_t)w * 6 * v.x_max + 6 * eo + j) * XL_LD;
                    if (res) x = v.xl_out[((size_t)b * v.B + w) * 6 * v.x_max + 6 * eo + j];
                    else if (c < 21) x = U[c];                                     // m, pose of m + 1
                    else if (c >= 30 && c < 36) x = U[21 + c - 30];                // pose of m + 2
                    else if (tc >= 0 && f_src[tc] < 0) x = U[27 + 6 * (-1 - f_src[tc]) + cc];   // (old far ends keep their order: tc = that end)
                } else {
                    const double* F = v.x_out + (((size_t)b * v.B + w) * v.x_max + f_src[t]) * BTW_OUT;
                    if (res) x = F[j];
                    else if (c < 6) x = F[6 + j * 6 + c];
                    else if (tc == t) x = F[42 + j * 6 + cc];
                }
            }
            Wf[e] = x;
        }
        __syncthreads();
        // a far end at m + 3: its columns join the pose columns of that keyframe (36..41)
        for (int e = lane; e < R * 6; e += 256) {
            const int row = e / 6, c = e - row * 6;
            double add = 0.0;
            for (int t = 0; t < T; t++)
                if (f_fold[t]) { add += Wf[row * FC + 42 + 6 * t + c]; Wf[row * FC + 42 + 6 * t + c] = 0.0; }
            Wf[row * FC + 36 + c] += add;
        }
        __syncthreads();
        for (int e = lane; e < 42 * 43; e += 256) {
            const int i = e / 43, j = e - i * 43;
            double sum = 0.0;
            for (int r = 0; r < R; r++) sum = fma(Wf[r * FC + i], j < 42 ? Wf[r * FC + j] : Wf[r * FC + NC - 1], sum);
            if (j < 42) A[i * 43 + j] += sum; else bv[i] += sum;
        }
        for (int e = lane; e < R * FC; e += 256) {
            const int p = e / FC, c = e - p * FC;
            double sum = 0.0;
            if (c < NC) for (int r = 0; r < R; r++) sum = fma(Wf[r * FC + 42 + p], Wf[r * FC + c], sum);
            E[e] = sum;
        }
        __syncthreads();
    }
    int bad = 0;
    for (int c = 0; c < 15; c++) {
        const double d = A[c * 43 + c];
        if (!(d > 0.0)) bad = 1;
        const double inv = 1.0 / d;
        __syncthreads();
        // rank-1 update of the trailing (41-c)x(41-c) block and of b, entry-parallel; column c and
        // b[c] are only read in this step, so one pass is hazard-free
        const int m = 41 - c;
        for (int e = lane; e < m * m + m; e += 256) {
            const bool is_b = e >= m * m;
            const int i = c + 1 + (is_b ? e - m * m : e / m), j = is_b ? -1 : c + 1 + (e - (e / m) * m);
            const double u = A[i * 43 + c] * inv * (is_b ? bv[c] : A[j * 43 + c]);
            if (is_b) bv[i] -= u; else A[i * 43 + j] -= u;
        }
        if constexpr (FAR) {
            // the far-end rows: E[p][j] -= E[p][c] / d * (row c of the whole system)[j]; row c is A[c][.] over the 42 columns,
            // E[.][c] over the far-end columns (symmetry), bv[c] for the right-hand side
            const int R = 6 * f_T, NC = 42 + R + 1;
            for (int e = lane; e < R * FC; e += 256) {
                const int p = e / FC, j = e - p * FC;
                if (j <= c || j >= NC) continue;
                const double rc = j < 42 ? A[c * 43 + j] : (j < NC - 1 ? E[(j - 42) * FC + c] : bv[c]);
                E[e] -= E[p * FC + c] * inv * rc;
            }
        }
        __syncthreads();
    }
    if constexpr (FAR) {
        // ---- split the marginal over (n, far ends) into the prior on n and six rows per live far end
        const int T = f_T;
        if (lane == 0) { int L = 0; for (int t = 0; t < T; t++) if (!f_fold[t]) f_src[L++] = t; f_live = L; }    // (f_src: now the live far ends, in order)
        __syncthreads();
        const int Lv = f_live, RL = 6 * Lv;
        auto pr = [&](int q) { return 6 * f_src[q / 6] + q % 6; };              // live row / column q -> row of E (column 42 + that)
        // S_bb (live x live) -> Wf[q][q2] (row stride FC), then Cholesky in place, right-looking, entry-parallel
        for (int e = lane; e < RL * RL; e += 256) { const int q = e / RL, q2 = e - q * RL; Wf[q * FC + q2] = 0.5 * (E[pr(q) * FC + 42 + pr(q2)] + E[pr(q2) * FC + 42 + pr(q)]); }
        __syncthreads();
        for (int c = 0; c < RL; c++) {
            const double d = Wf[c * FC + c];
            if (!(d > 0.0)) bad |= 4;                // (status bit 4: the far ends' block of the marginal is not positive definite)
#ifdef VF_DEBUG_FAR
            if (!(d > 0.0) && lane == 0) printf("[far] window %d lo %d: S_bb pivot %d of %d = %.6e (T %d, live %d; far ends %d %d %d %d %d %d %d %d; fold %d %d %d %d %d %d %d %d; first diag %.3e)\n", w, lo, c, RL, d, T, Lv,
                                                f_kb[0], f_kb[1], f_kb[2], f_kb[3], f_kb[4], f_kb[5], f_kb[6], f_kb[7], f_fold[0], f_fold[1], f_fold[2], f_fold[3], f_fold[4], f_fold[5], f_fold[6], f_fold[7], Wf[0]);
#endif
            const double sd = sqrt(d > 0.0 ? d : 1.0);
            __syncthreads();
            for (int q = c + lane; q < RL; q += 256) Wf[q * FC + c] = q == c ? sd : Wf[q * FC + c] / sd;
            __syncthreads();
            const int rem = RL - 1 - c;
            for (int e = lane; e < rem * rem; e += 256) {
                const int q = c + 1 + e / rem, q2 = c + 1 + (e - (e / rem) * rem);
                if (q2 <= q) Wf[q * FC + q2] -= Wf[q * FC + c] * Wf[q2 * FC + c];
            }
            __syncthreads();
        }
        // X = L^-1 [S_bn | eta_b]: thread per column (27 + 1), into Wf[q][RL + col]
        if (lane < 28) {
            const int src = lane < 27 ? 15 + lane : 42 + 6 * T;                 // column of E: n's 27 (A's 15..41), the right-hand side
            for (int q = 0; q < RL; q++) {
                double a = E[pr(q) * FC + src];
                for (int k = 0; k < q; k++) a = fma(-Wf[q * FC + k], Wf[k * FC + RL + lane], a);
                Wf[q * FC + RL + lane] = a / Wf[q * FC + q];
            }
        }
        __syncthreads();
        // the prior on n loses what the rows now carry
        for (int e = lane; e < 27 * 28; e += 256) {
            const int i = e / 28, j = e - i * 28;
            double sum = 0.0;
            for (int q = 0; q < RL; q++) sum = fma(Wf[q * FC + RL + i], Wf[q * FC + RL + j], sum);
            if (j < 27) A[(15 + i) * 43 + 15 + j] -= sum; else bv[15 + i] -= sum;
        }
        // the rows: [X_n | L^T] over [n: 27][live far ends: 6 each], residual x_eta -- both buffers, the linearisation point
        // being the current states from here on
        for (int e = lane; e < RL * XL_LD; e += 256) {
            const int q = e / XL_LD, c = e - q * XL_LD;
            double x = 0.0;
            if (c < 27) x = Wf[q * FC + RL + c];
            else if (c - 27 < RL && c - 27 >= q) x = Wf[(c - 27) * FC + q];     // L^T: upper triangle
            v.xl_U[((size_t)w * 6 * v.x_max + q) * XL_LD + c] = x;
        }
        for (int q = lane; q < 6 * v.x_max; q += 256) {
            const double x = q < RL ? Wf[q * FC + RL + 27] : 0.0;
            v.xl_r0[(size_t)w * 6 * v.x_max + q] = x;
            v.xl_out[((size_t)0 * v.B + w) * 6 * v.x_max + q] = x;
            v.xl_out[((size_t)1 * v.B + w) * 6 * v.x_max + q] = x;
        }
        __syncthreads();                                 // (xl_b is read through f_kb, written below)
        if (lane < Lv * 7) {
            const int q = lane / 7, c = lane - 7 * q;
            v.xl_bx[((size_t)w * v.x_max + q) * 7 + c] = XS(b, c, (long)w * v.M + f_kb[f_src[q]]);
        }
        if (lane >= 64 && lane < 64 + Lv) v.xl_b[w * v.x_max + lane - 64] = f_kb[f_src[lane - 64]];
        if (lane == 128) v.xl_n[w] = Lv;
        if (lane >= 192 && lane < 192 + v.x_max) {       // the nonlinear entries that have just become linear (the host re-sends the list without them)
            const int i = w * v.x_max + lane - 192;
            if (v.x_a[i] == lo && v.x_b[i] - lo > 3 && v.x_b[i] < hi) v.x_a[i] = -1;
        }
        __syncthreads();
    }
    // Gauge floor (vf_engine_opts.gauge_floor).  Every factor of the window is invariant under a global translation and a
    // rotation about gravity; what the window knows about those four directions is G^T L G of this prior alone -- the memory
    // of the anchor prior, which decays with every marginalisation until it is below the rounding of the 1e9-scale entries
    // beside it (cond(H) 5e12 after 100 updates of a 200-keyframe window, 1e16 after 2 000, indefinite after 3 000: LM trials
    // rejected at random, then failed solves).  Eigenvalues of G^T L G that have fallen below the floor are lifted back to it
    // (the default keeps the window's softest eigenvalue at 3e-4, two orders above the rounding of its largest entries: for
    // 1 000 keyframes an information of 0.1 on WHERE the window is, a 3 m sigma -- nothing the factors can see).
    if (v.gauge_floor > 0.0) {
        __shared__ double Gq[27 * 4], Tq[27 * 4], Mq[16], Vq[16], lift[4];
        for (int e = lane; e < 729; e += 256) {          // the symmetric part, in place
            const int i = e / 27, j = e - i * 27;
            if (j > i) { const double sy = 0.5 * (A[(15 + i) * 43 + 15 + j] + A[(15 + j) * 43 + 15 + i]); A[(15 + i) * 43 + 15 + j] = sy; A[(15 + j) * 43 + 15 + i] = sy; }
        }
        for (int e = lane; e < 108; e += 256) Gq[e] = 0.0;
        __syncthreads();
        if (lane < 3) {                                  // lane j: the rows of kept keyframe j
            const int j = lane, o = j == 0 ? 0 : 15 + 6 * (j - 1);
            const State s0 = load_state(v, b, g0 + 1), sj = load_state(v, b, g0 + 1 + j);
            const M3 R = qrot(sj.q);
            const double gn = sqrt(v.grav[0] * v.grav[0] + v.grav[1] * v.grav[1] + v.grav[2] * v.grav[2]);
            const V3 ez = gn > 0.0 ? v3(-v.grav[0] / gn, -v.grav[1] / gn, -v.grav[2] / gn) : v3(0.0, 0.0, 1.0);
            const V3 dt = sj.t - s0.t;
            const V3 lever = v3(ez.y * dt.z - ez.z * dt.y, ez.z * dt.x - ez.x * dt.z, ez.x * dt.y - ez.y * dt.x);
            for (int c = 0; c < 3; c++) {
                for (int a = 0; a < 3; a++) Gq[(o + 3 + c) * 4 + a] = R.a[a * 3 + c];
                Gq[(o + c) * 4 + 3] = R.a[0 * 3 + c] * ez.x + R.a[1 * 3 + c] * ez.y + R.a[2 * 3 + c] * ez.z;
                Gq[(o + 3 + c) * 4 + 3] = R.a[0 * 3 + c] * lever.x + R.a[1 * 3 + c] * lever.y + R.a[2 * 3 + c] * lever.z;
            }
            if (j == 0) {
                Gq[6 * 4 + 3] = ez.y * sj.vel.z - ez.z * sj.vel.y;
                Gq[7 * 4 + 3] = ez.z * sj.vel.x - ez.x * sj.vel.z;
                Gq[8 * 4 + 3] = ez.x * sj.vel.y - ez.y * sj.vel.x;
            }
        }
        __syncthreads();
        if (lane == 0) {                                 // modified Gram-Schmidt, 4 columns of 27
            for (int a = 0; a < 4; a++) {
                for (int bb = 0; bb < a; bb++) {
                    double s = 0.0;
                    for (int i = 0; i < 27; i++) s += Gq[i * 4 + a] * Gq[i * 4 + bb];
                    for (int i = 0; i < 27; i++) Gq[i * 4 + a] -= s * Gq[i * 4 + bb];
                }
                double nn = 0.0;
                for (int i = 0; i < 27; i++) nn += Gq[i * 4 + a] * Gq[i * 4 + a];
                nn = sqrt(nn);
                for (int i = 0; i < 27; i++) Gq[i * 4 + a] /= nn;
            }
        }
        __syncthreads();
        if (lane < 108) {
            const int i = lane >> 2, a = lane & 3;
            double s = 0.0;
            for (int j = 0; j < 27; j++) s += A[(15 + i) * 43 + 15 + j] * Gq[j * 4 + a];
            Tq[lane] = s;
        }
        __syncthreads();
        if (lane < 16) {
            const int a = lane >> 2, bb = lane & 3;
            double s = 0.0;
            for (int i = 0; i < 27; i++) s += Gq[i * 4 + a] * Tq[i * 4 + bb];
            Mq[lane] = s;
        }
        __syncthreads();
        if (lane == 0) {                                 // cyclic Jacobi on the 4 x 4 (10 sweeps)
            for (int a = 0; a < 4; a++)
                for (int bb = a + 1; bb < 4; bb++) { const double sy = 0.5 * (Mq[a * 4 + bb] + Mq[bb * 4 + a]); Mq[a * 4 + bb] = sy; Mq[bb * 4 + a] = sy; }
            for (int i = 0; i < 16; i++) Vq[i] = (i % 5 == 0) ? 1.0 : 0.0;
            for (int sweep = 0; sweep < 10; sweep++)
                for (int p = 0; p < 3; p++)
                    for (int q = p + 1; q < 4; q++) {
                        const double apq = Mq[p * 4 + q];
                        if (apq == 0.0) continue;
                        const double theta = (Mq[q * 4 + q] - Mq[p * 4 + p]) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                        for (int kk = 0; kk < 4; kk++) {
                            const double mkp = Mq[kk * 4 + p], mkq = Mq[kk * 4 + q];
                            Mq[kk * 4 + p] = c * mkp - sn * mkq;
                            Mq[kk * 4 + q] = sn * mkp + c * mkq;
                            const double vkp = Vq[kk * 4 + p], vkq = Vq[kk * 4 + q];
                            Vq[kk * 4 + p] = c * vkp - sn * vkq;
                            Vq[kk * 4 + q] = sn * vkp + c * vkq;
                        }
                        for (int kk = 0; kk < 4; kk++) {
                            const double mpk = Mq[p * 4 + kk], mqk = Mq[q * 4 + kk];
                            Mq[p * 4 + kk] = c * mpk - sn * mqk;
                            Mq[q * 4 + kk] = sn * mpk + c * mqk;
                        }
                    }
            // (the floor is set on the WINDOW's eigenvalue: a unit gauge vector of an n-keyframe window has 3 / n of its weight
            // on the three keyframes this prior touches)
            const double floor_p = v.gauge_floor * (double)(v.hi[w] - v.lo[w]) / 3.0;
            for (int e = 0; e < 4; e++) { const double l = floor_p - Mq[e * 4 + e]; lift[e] = l > 0.0 ? l : 0.0; }
        }
        __syncthreads();
        if (lane < 108) {                                // q_e = G V[:, e]
            const int i = lane >> 2, e = lane & 3;
            double s = 0.0;
            for (int a = 0; a < 4; a++) s += Gq[i * 4 + a] * Vq[a * 4 + e];
            Tq[lane] = s;
        }
        __syncthreads();
        for (int e = lane; e < 729; e += 256) {
            const int i = e / 27, j = e - i * 27;
            double add = 0.0;
            for (int q = 0; q < 4; q++) add += lift[q] * Tq[i * 4 + q] * Tq[j * 4 + q];
            A[(15 + i) * 43 + 15 + j] += add;
        }
        __syncthreads();
    }
    // new marginal prior on [m+1:15][m+2 pose][m+3 pose] = rows 15..41, relinearised at the current states
    if (!FAR && stash) {
        double* st = stash + (size_t)w * MARG_STASH;
        for (int e = lane; e < 729 + 27; e += 256) {
            if (e < 729) { const int i = e / 27, j = e - i * 27; st[e] = 0.5 * (A[(15 + i) * 43 + 15 + j] + A[(15 + j) * 43 + 15 + i]); }
            else st[e] = bv[15 + e - 729];
        }
        if (lane < 48) { const int j = lane / 16, c = lane - j * 16; st[756 + lane] = XS(b, c, g0 + 1 + j); }
        if (lane == 0 && bad) atomicOr(status, bad & 4 ? 4 : 1);
        return;
    }
    for (int e = lane; e < 729 + 27; e += 256) {
        if (e < 729) { const int i = e / 27, j = e - i * 27; v.mp_L[(size_t)w * 729 + e] = 0.5 * (A[(15 + i) * 43 + 15 + j] + A[(15 + j) * 43 + 15 + i]); }
        else {
            // also the "linearised" form at the new linearisation point (d = 0: gradient eta, cost 0),
            // so that a second marginalisation can follow without an intervening linearise
            const double et = bv[15 + e - 729];
            v.mp_eta[(size_t)w * 27 + e - 729] = et;
            v.mp_out[((size_t)0 * v.B + w) * 28 + e - 729] = et;
            v.mp_out[((size_t)1 * v.B + w) * 28 + e - 729] = et;
        }
    }
    if (lane == 0) { v.mp_out[((size_t)0 * v.B + w) * 28 + 27] = 0.0; v.mp_out[((size_t)1 * v.B + w) * 28 + 27] = 0.0; }
    if (lane < 48) { const int j = lane / 16, c = lane - j * 16; v.mp_x[(size_t)w * 48 + lane] = XS(b, c, g0 + 1 + j); }
    if (lane == 0) {
        v.mp_on[w] = 1;
        v.prior_k[w] = -1;
        if (bad) atomicOr(status, bad & 4 ? 4 : 1);
    }
}

// what k_marginalize (stash != null) computed ahead of time becomes the window's marginal prior, exactly as the kernel itself
// would have left it
__global__ void __launch_bounds__(256) k_marg_commit(View v, const double* __restrict__ stash) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const double* st = stash + (size_t)w * MARG_STASH;
    for (int e = lane; e < 729 + 27; e += 256) {
        if (e < 729) v.mp_L[(size_t)w * 729 + e] = st[e];
        else {
            const double et = st[e];
            v.mp_eta[(size_t)w * 27 + e - 729] = et;
            v.mp_out[((size_t)0 * v.B + w) * 28 + e - 729] = et;
            v.mp_out[((size_t)1 * v.B + w) * 28 + e - 729] = et;
        }
    }
    if (lane == 0) { v.mp_out[((size_t)0 * v.B + w) * 28 + 27] = 0.0; v.mp_out[((size_t)1 * v.B + w) * 28 + 27] = 0.0; }
    if (lane < 48) v.mp_x[(size_t)w * 48 + lane] = st[756 + lane];
    if (lane == 0) {
        v.mp_on[w] = 1;
        v.prior_k[w] = -1;
    }
}

// fixed-lag slide by one keyframe: hi += 1 (predict the new state), lo += 1, and either re-anchor
// the diagonal prior on the new oldest keyframe (reanchor = 1) or keep the marginal prior that
// k_marginalize just produced (reanchor = 0)
__global__ void k_slide(View v, const double* sigma15, int reanchor) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= v.B) return;
    const int b = v.sel[w];
    const int hi = v.hi[w], lo = v.lo[w];
    if (hi >= v.M) return;  // out of slots: caller must compact
    const long gnew = (long)w * v.M + hi;
    store_state(v, b, gnew, predict_state(v, load_state(v, b, gnew - 1), gnew));
    v.hi[w] = hi + 1;
    v.lo[w] = lo + 1;
    if (!reanchor) return;
    const long ganchor = (long)w * v.M + lo + 1;
    v.prior_k[w] = lo + 1;
    v.mp_on[w] = 0;
    double* pin = v.prior_in + (size_t)w * PRIOR_IN;
    for (int c = 0; c < 16; c++) pin[c] = XS(b, c, ganchor);
    for (int c = 0; c < 15; c++) pin[16 + c] = sigma15[c];
}

// ------------------------------------------------------------------------------------ compaction
// Shift the live part of every window down by `shift` slots (a multiple of 64, so AoSoA tiles
// move whole).  Source and destination ranges of a window may overlap, hence the staging copy
// through `tmp` (sized for one array at a time by the host).  Element e of a per-window array
// with `per` doubles (or ints) per slot-tile unit.
__global__ void k_shift_copy(const double* __restrict__ src, double* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void k_shift_btw_a(int* a, long G, int M, int shift) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const int k = (int)(g % M);
    // after the move, slot k holds what was in slot k + shift; indices of `a` drop by shift
    const int v0 = a[g];
    (void)k;
    a[g] = v0 >= shift ? v0 - shift : -1;
}

// ------------------------------------------------------------------------------------ staging
__global__ void k_scatter(const double* aos, double* aosoa, long g0, long n, int nf) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nf) return;
    const long rec = i / nf;
    const int f = (int)(i - rec * nf);
    const long g = g0 + rec;
    aosoa[((size_t)(g >> 6) * nf + f) * TILE + (g & 63)] = aos[i];
}
__global__ void k_gather(const double* aosoa, double* aos, long g0, long n, int nf) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nf) return;
    const long rec = i / nf;
    const int f = (int)(i - rec * nf);
    const long g = g0 + rec;
    aos[i] = aosoa[((size_t)(g >> 6) * nf + f) * TILE + (g & 63)];
}
__global__ void k_gather_imu_lin(View v, int b, long g0, long n, double* aos) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * IMU_OUT) return;
    const long rec = i / IMU_OUT;
    const int f = (int)(i - rec * IMU_OUT);
    const long g = g0 + rec;
    if (f < IMU_R) aos[i] = v.imu_r[((size_t)b * (size_t)(v.G >> 6) + (size_t)(g >> 6)) * IMU_R * TILE + (size_t)f * TILE + (g & 63)];
    else aos[i] = jstream_entry(v.imu_j + (size_t)b * (size_t)(v.G >> JT_LOG) * JT_STRIDE, g, (f - 15) / 30, (f - 15) % 30);
}
__global__ void k_scatter_states(const double* aos, double* x, long G, int buf, long g0, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 16) return;
    const long rec = i / 16;
    const int c = (int)(i - rec * 16);
    x[((size_t)buf * 16 + c) * (size_t)G + g0 + rec] = aos[i];
}
__global__ void k_gather_states(const double* x, double* aos, long G, const int* sel, int M, int which, long g0, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 16) return;
    const long rec = i / 16;
    const int c = (int)(i - rec * 16);
    const long g = g0 + rec;
    const int buf = sel[g / M] ^ which;
    aos[i] = x[((size_t)buf * 16 + c) * (size_t)G + g];
}

// ------------------------------------------------------------------------------------ launchers
static inline unsigned nblk(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

void launch_preintegrate(const View& v, long g0, int n, const int* off, const double* steps, const double* bhat6,
                         const ImuCov& prm, int* status, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_preintegrate_t<false>, dim3(n), dim3(256), 0, s, v, g0, n, off, steps, bhat6,
                                  (const int*)nullptr, (const double*)nullptr, prm, status);
}
void launch_ingest_tail(const View& v, const int* off, const double* steps, const int* tail_a, const double* tail_btw,
                        const ImuCov& prm, int* status, hipStream_t s) {
    hipLaunchKernelGGL(k_preintegrate_t<true>, dim3(v.B), dim3(256), 0, s, v, 0L, v.B, off, steps, (const double*)nullptr, tail_a,
                       tail_btw, prm, status);
}
void launch_linearize_extra(const View& v, int which, hipStream_t s) {
    if (v.x_max > 0) hipLaunchKernelGGL(k_linearize_extra, dim3(nblk((long)v.B * v.x_max, 64)), dim3(64), 0, s, v, which);
}
void launch_extra_gradient(const View& v, hipStream_t s) {
    if (v.x_max > 0) hipLaunchKernelGGL(k_extra_gradient, dim3(v.B), dim3(64), 0, s, v);
}
void launch_extra_rhs(const View& v, int slot, int row, double* gtmp, hipStream_t s) {
    hipLaunchKernelGGL(k_extra_rhs, dim3(nblk(v.B, 64)), dim3(64), 0, s, v, slot, row, gtmp);
}
void launch_extra_combine(const View& v, const double* Zm, size_t zstride, int slots, hipStream_t s) {
    hipLaunchKernelGGL(k_extra_combine, dim3(v.B), dim3(256), 0, s, v, Zm, zstride, slots);
}
void launch_linearize_all(const View& v, int which, hipStream_t s) {
    const int nb_imu = (int)nblk(v.G, VF_K1_BLOCK), nb_btw = nb_imu, nb_pri = (int)nblk(v.B, VF_K1_BLOCK);
    if (v.sh_G > 1) hipLaunchKernelGGL(k_linearize_all<true>, dim3(nb_imu + nb_btw + nb_pri), dim3(VF_K1_BLOCK), 0, s, v, which, nb_imu, nb_btw);
    else hipLaunchKernelGGL(k_linearize_all<false>, dim3(nb_imu + nb_btw + nb_pri), dim3(VF_K1_BLOCK), 0, s, v, which, nb_imu, nb_btw);
}
void launch_linearize_imu(const View& v, int which, hipStream_t s) {
    hipLaunchKernelGGL(k_linearize_imu, dim3(nblk(v.G, VF_K1_BLOCK)), dim3(VF_K1_BLOCK), 0, s, v, which);
}
void launch_linearize_between(const View& v, int which, hipStream_t s) {
    hipLaunchKernelGGL(k_linearize_between, dim3(nblk(v.G, 256)), dim3(256), 0, s, v, which);
}
void launch_linearize_between_prior(const View& v, int which, hipStream_t s) {
    const int nb_pri = (int)nblk(v.B, 64);
    hipLaunchKernelGGL(k_linearize_between_prior, dim3(nblk(v.G, 256) + nb_pri), dim3(256), 0, s, v, which, nb_pri);
}
void launch_linearize_tail(const View& v, int nslid, hipStream_t s) {
    hipLaunchKernelGGL(k_linearize_tail, dim3(2 * v.B + nblk(v.B, 64)), dim3(VF_K1_BLOCK), 0, s, v, nslid);
}
void launch_linearize_prior(const View& v, int which, hipStream_t s) {
    hipLaunchKernelGGL(k_linearize_prior, dim3(nblk(v.B, 64)), dim3(64), 0, s, v, which);
}
// K3 for the partitioned half of a hybrid solve whose sweep half assembles its own rows (see k_assemble)
void launch_assemble_for_partitioned(const View& v, hipStream_t s) {
    View a = v;
    a.gate = 2;
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)(v.M / AT), (unsigned)v.B), dim3(K3_NT), 0, s, a);
}
void launch_assemble(const View& v, hipStream_t s) {
    // (persistent forms were measured slower: one workgroup per CU with the next tile's loads in flight 5.1 ms, a plain
    // tile loop on 1024-2048 workgroups 4.0-4.3 ms, against 2.7 ms for one workgroup per tile -- stores count in vmcnt
    // on this ISA, so a loop waits for its own H stores before it can use the next tile's loads)
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)(v.M / AT), (unsigned)v.B), dim3(K3_NT), 0, s, v);
}
void launch_assemble_window(const View& v, int window, hipStream_t s) {
    // (gate = 2: K3 ignores `fresh`; stop_on = 0: a window the termination rule has finished is assembled too)
    View a = v;
    a.gate = 2;
    a.gate_T = 1 << 30;          // ... and gated_off() lets a gate-2 launch run while n_active <= gate_T
    a.stop_on = 0;
    a.w_first = window;
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)(v.M / AT), 1u), dim3(K3_NT), 0, s, a);
}
void launch_partitioned_local(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_chunk_forward, dim3((unsigned)v.B * (unsigned)v.P), dim3(128), 0, s, v);
}
void launch_partitioned_global(const View& v, hipStream_t s) {
    const unsigned nb = (unsigned)v.B * (unsigned)v.P;
    hipLaunchKernelGGL(k_sep_solve, dim3(v.B), dim3(256), 0, s, v);
    hipLaunchKernelGGL(k_chunk_rhs, dim3(nb), dim3(256), 0, s, v);
    hipLaunchKernelGGL(k_chunk_back, dim3(nb), dim3(64), 0, s, v);
}
void launch_partitioned_solve(const View& v, hipStream_t s) {
    launch_partitioned_local(v, s);
    launch_partitioned_global(v, s);
}
// time-sharded windows, after the back substitution: zero the increments of the keyframes this rank does not own and put
// this rank's solve-failure flags behind them, so that ONE all-reduce (sum) leaves every rank with all increments and
// with the number of ranks whose elimination failed
__global__ void __launch_bounds__(256) k_mask_delta(View v) {
    const long gk = (long)blockIdx.x * 256 + threadIdx.x;
    if (gk < v.B) v.delta[(size_t)v.G * 15 + gk] = v.fail[gk] ? 1.0 : 0.0;
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    const int lo = v.lo[w];
    if (k < lo || k >= v.hi[w]) return;
    int klo, khi;
    own_range(v, w, klo, khi);
    if (k - lo >= klo && k - lo < khi) return;
#pragma unroll
    for (int a = 0; a < 15; a++) v.delta[(size_t)gk * 15 + a] = 0.0;
}
void launch_mask_delta(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_mask_delta, dim3(nblk(v.G, 256)), dim3(256), 0, s, v);
}
static void launch_asm2(const View& v, hipStream_t s) {
    // one launch per 1 024 windows -- the workgroups the part holds at once (4 per CU).  As ONE grid of 2 048 workgroups
    // the second thousand, dispatched one by one into the slots the first leaves, ran 1.7x slower than the first
    // (9.0 ms against 7.1 for the two launches; no such effect on the one-wave kernels)
    for (int w0 = 0; w0 < v.B; w0 += VF_ASM2_CHUNK) {
        const unsigned nb = (unsigned)(v.B - w0 < VF_ASM2_CHUNK ? v.B - w0 : VF_ASM2_CHUNK);
        if (VF_ASM2_ROLES) (void)hipMemsetAsync(v.place, 0, PLACE_CELLS * sizeof(unsigned), s);      // the per-CU claims of this launch
        hipLaunchKernelGGL(k_band_forward_asm2, dim3(nb), dim3(128), 0, s, v, w0);
    }
}
void launch_band_solve(const View& v, hipStream_t s) {
    if (v.P >= 2) { launch_partitioned_solve(v, s); return; }
    // few windows: two waves per window from both ends (latency); many: one wave per window (throughput)
    if (v.B <= v.tw_max) hipLaunchKernelGGL(k_band_solve_tw, dim3(v.B), dim3(128), 0, s, v);
    else if (asm_in_solve(v)) {
        if (v.asm_waves == 2) {
            launch_asm2(v, s);
        }
        else hipLaunchKernelGGL(k_band_forward_asm, dim3(v.B), dim3(64), 0, s, v);
        hipLaunchKernelGGL(k_band_backward, dim3(v.B), dim3(64), 0, s, v);
    } else if (v.split_min > 0 && v.B >= v.split_min) {
        hipLaunchKernelGGL(k_band_forward, dim3(v.B), dim3(64), 0, s, v);
        hipLaunchKernelGGL(k_band_backward, dim3(v.B), dim3(64), 0, s, v);
    } else hipLaunchKernelGGL(k_band_solve, dim3(v.B), dim3(64), 0, s, v);
}
// windows of the running solve that still take LM trials (termination rule on)
__global__ void __launch_bounds__(1024) k_count_active(View v) {
    // ... and their compacted list, in window order (a block-wide inclusive scan per 1024 windows)
    __shared__ int scan[1024];
    __shared__ int base;
    const int tid = threadIdx.x;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int w0 = 0; w0 < v.B; w0 += 1024) {
        const int w = w0 + tid;
        const int on = (w < v.B && v.hi[w] > v.lo[w] && !v.done[w]) ? 1 : 0;
        scan[tid] = on;
        __syncthreads();
        for (int st = 1; st < 1024; st <<= 1) {
            const int add = tid >= st ? scan[tid - st] : 0;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        if (on && v.act) v.act[base + scan[tid] - 1] = w;
        __syncthreads();
        if (tid == 0) base += scan[1023];
        __syncthreads();
    }
    if (tid == 0) *v.n_active = base;
}
void launch_count_active(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_count_active, dim3(1), dim3(1024), 0, s, v);
}
void launch_band_solve_hybrid(const View& v, const View& vp, hipStream_t s) {
    View a = v, b = vp;
    a.gate = 1;
    b.gate = 2;
    // (a.act, when the engine has allocated it: the sweeps visit the active windows first)
    if (asm_in_hybrid(v)) {
        // (two waves per window here too, round 5: with the compacted list, the eliminator's priority and -- what had made it
        // 5 % slower than one wave in round 4 and 21 % slower than itself in the headline's launch -- the placement kernel in
        // front of it, launch_asm2)
        if (a.asm_waves == 2 && a.act) launch_asm2(a, s);
        else
        hipLaunchKernelGGL(k_band_forward_asm, dim3(a.B), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_band_backward, dim3(a.B), dim3(64), 0, s, a);
    } else if (a.split_min > 0 && a.B >= a.split_min) {
        hipLaunchKernelGGL(k_band_forward, dim3(a.B), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_band_backward, dim3(a.B), dim3(64), 0, s, a);
    } else hipLaunchKernelGGL(k_band_solve, dim3(a.B), dim3(64), 0, s, a);
    launch_partitioned_solve(b, s);
}
void launch_retract(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_retract, dim3(nblk(v.G, 256)), dim3(256), 0, s, v);
}
void launch_reset_lambda(const View& v, const double* lambda0, hipStream_t s) {
    hipLaunchKernelGGL(k_reset_lambda, dim3(nblk(v.B, 256)), dim3(256), 0, s, v, lambda0);
}
void launch_close_excursions(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_close_excursion, dim3(v.B), dim3(v.B <= 64 ? 1024 : 256), 0, s, v);
}
void launch_decide(const View& v, int init, hipStream_t s) {
    hipLaunchKernelGGL(k_decide, dim3(v.B), dim3(v.B <= 64 ? 1024 : 256), 0, s, v, init);
}
void launch_predict(const View& v, int window, int k0, int n, int from_trial, hipStream_t s) {
    if (window >= 0) hipLaunchKernelGGL(k_predict, dim3(1), dim3(1), 0, s, v, window, k0, n, from_trial);
    else hipLaunchKernelGGL(k_predict, dim3(nblk(v.B, 64)), dim3(64), 0, s, v, window, k0, n, from_trial);
}
void launch_relinearize(const View& v, double threshold, hipStream_t s) {
    hipLaunchKernelGGL(k_relinearize, dim3(nblk(v.G, 256)), dim3(256), 0, s, v, threshold);
}
__global__ void k_set_range(View v, int window, int lo, int hi) {
    if (lo >= 0) v.lo[window] = lo;
    v.hi[window] = hi;
}
__global__ void k_bump_lo(View v) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < v.B) v.lo[w] += 1;
}
__global__ void __launch_bounds__(64) k_put_between(View v, long g, int a, BtwArg rec) {
    const int f = threadIdx.x;
    if (f < BTW_IN) v.btw_in[(size_t)(g >> 6) * BTW_IN * TILE + (size_t)f * TILE + (g & 63)] = rec.r[f];
    if (f == 0) v.btw_a[g] = a;
}
__global__ void __launch_bounds__(64) k_read_result(View v, int window, int slot, int which, int* sticky, SolveResult* out) {
    const int t = threadIdx.x;
    const int b = v.sel[window] ^ which;
    if (t < 16) out->state[t] = XS(b, t, (long)window * v.M + slot);
    if (t == 16) { out->cost = v.cost[window]; out->n_acc = v.n_acc[window]; out->n_rej = v.n_rej[window]; out->n_fail = v.n_fail[window]; }
    if (t == 17) { out->sticky[0] = sticky[0]; out->sticky[1] = sticky[1]; sticky[0] = 0; sticky[1] = 0; }
}
void launch_set_range(const View& v, int window, int lo, int hi, hipStream_t s) {
    hipLaunchKernelGGL(k_set_range, dim3(1), dim3(1), 0, s, v, window, lo, hi);
}
void launch_bump_lo(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_bump_lo, dim3(nblk(v.B, 64)), dim3(64), 0, s, v);
}
void launch_put_between(const View& v, long g, int a, const BtwArg& rec, hipStream_t s) {
    hipLaunchKernelGGL(k_put_between, dim3(1), dim3(64), 0, s, v, g, a, rec);
}
void launch_read_result(const View& v, int window, int slot, int which, int* sticky, SolveResult* out, hipStream_t s) {
    hipLaunchKernelGGL(k_read_result, dim3(1), dim3(64), 0, s, v, window, slot, which, sticky, out);
}
void launch_inc_begin(const View& v, double threshold, int appended, int invalid, hipStream_t s) {
    hipLaunchKernelGGL(k_inc_begin, dim3((unsigned)nblk(v.M, 256), (unsigned)v.B), dim3(256), 0, s, v, threshold, appended, invalid);
}
void launch_inc_solve(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_inc_forward, dim3(v.B), dim3(64), 0, s, v);
    hipLaunchKernelGGL(k_inc_backward, dim3(v.B), dim3(64), 0, s, v);
}
void launch_inc_retract(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_inc_retract, dim3((unsigned)nblk(v.M, 256), (unsigned)v.B), dim3(256), 0, s, v);
}
void launch_slide(const View& v, const double* sigma15_dev, int reanchor, hipStream_t s) {
    hipLaunchKernelGGL(k_slide, dim3(nblk(v.B, 64)), dim3(64), 0, s, v, sigma15_dev, reanchor);
}
void launch_marginalize_ahead(const View& v, int* status, double* stash, hipStream_t s) {
    hipLaunchKernelGGL(k_marginalize<false>, dim3(v.B), dim3(256), 0, s, v, status, stash);
}
void launch_marg_commit(const View& v, const double* stash, hipStream_t s) {
    hipLaunchKernelGGL(k_marg_commit, dim3(v.B), dim3(256), 0, s, v, stash);
}
void launch_marginalize(const View& v, int* status, hipStream_t s) {
    if (v.x_max > 0) hipLaunchKernelGGL(k_marginalize<true>, dim3(v.B), dim3(256), 0, s, v, status, (double*)nullptr);
    else hipLaunchKernelGGL(k_marginalize<false>, dim3(v.B), dim3(256), 0, s, v, status, (double*)nullptr);
}
void launch_shift_copy(const double* src, double* dst, long n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_shift_copy, dim3(nblk(n, 256)), dim3(256), 0, s, src, dst, n);
}
void launch_shift_btw_a(int* a, long G, int M, int shift, hipStream_t s) {
    hipLaunchKernelGGL(k_shift_btw_a, dim3(nblk(G, 256)), dim3(256), 0, s, a, G, M, shift);
}
void launch_scatter(const double* aos, double* aosoa, long g0, long n, int nf, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_scatter, dim3(nblk(n * nf, 256)), dim3(256), 0, s, aos, aosoa, g0, n, nf);
}
void launch_gather(const double* aosoa, double* aos, long g0, long n, int nf, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather, dim3(nblk(n * nf, 256)), dim3(256), 0, s, aosoa, aos, g0, n, nf);
}
void launch_gather_imu_lin(const View& v, int b, long g0, long n, double* aos, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather_imu_lin, dim3(nblk(n * IMU_OUT, 256)), dim3(256), 0, s, v, b, g0, n, aos);
}
void launch_scatter_states(const double* aos, double* x, long G, int buf, long g0, long n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_scatter_states, dim3(nblk(n * 16, 256)), dim3(256), 0, s, aos, x, G, buf, g0, n);
}
void launch_gather_states(const double* x, double* aos, long G, const int* sel, int M, int which, long g0, long n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather_states, dim3(nblk(n * 16, 256)), dim3(256), 0, s, x, aos, G, sel, M, which, g0, n);
}

}  // namespace vf
