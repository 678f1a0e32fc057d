// vf_kernels.hip -- gfx950 kernels of the smoother hot path.
//
//   K1 k_linearize_imu      CombinedImuFactor residual + whitened 15x30 Jacobian
//                           (replaces the linearisation ISAM2::update performs for the factor
//                           built at gtsam_fusion/src/gtsam_fusion/IMUManager.cpp:68-73)
//   K2 k_linearize_between  BetweenFactor<Pose3> residual + whitened 6x6 Jacobians
//                           (factor built at GraphManager.cpp:86)
//   K2b k_linearize_prior   the three priors of GraphManager.cpp:27-35
//   K3 k_assemble           block-banded J^T J / J^T r, owner-computes per keyframe
//   K4 k_band_solve         (H + lambda I) delta = -g, block-banded Cholesky, one wave/window
//   K5 k_retract, k_decide  x (+) delta, cost reduction, LM accept/reject
//   a2 k_predict, k_slide   PreintegrationBase::predict initial values (GraphManager.cpp:152-160)
//
// Mapping: one lane per factor / keyframe for K1-K3,K5 (HBM-bound, coalesced AoSoA tiles of 64:
// lane l of a wave reads word l of a 512-byte field row), one 64-lane wave per window for K4
// (latency-bound; panel factorisation in registers with v_readlane broadcasts, trailing
// window in LDS).
#include "vf_kernels.hpp"
#include "vf_math.hpp"
#include <cstdlib>

namespace vf {

#define XS(buf, c, gk) v.x[((size_t)(buf) * 16 + (c)) * (size_t)v.G + (size_t)(gk)]

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
// IMU preintegration, one lane per factor: PreintegratedCombinedMeasurements::integrateMeasurement
// over the factor's steps (mean, bias Jacobians, 15x15 covariance; GTSAM 4.0.x tangent form),
// then the noise model R = chol_upper(cov^-1) of the CombinedImuFactor built at
// gtsam_fusion/src/gtsam_fusion/IMUManager.cpp:68-73.  Runs once per factor (not per LM
// iteration); the 15x15 work uses per-lane local arrays.
__global__ void __launch_bounds__(64) k_preintegrate(View v, long g0, int n, const int* __restrict__ off,
                                                    const double* __restrict__ steps,
                                                    const double* __restrict__ bhat6, ImuCov prm, int* status) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= n) return;
    const double* bh = bhat6 + (size_t)f * 6;
    const V3 bacc = v3(bh[0], bh[1], bh[2]), bgyr = v3(bh[3], bh[4], bh[5]);
    V3 th = v3(0, 0, 0), pos = v3(0, 0, 0), vel = v3(0, 0, 0);
    double dtij = 0.0;
    double Hb[54], P[225], F[225], T[225];
    for (int i = 0; i < 54; i++) Hb[i] = 0.0;
    for (int i = 0; i < 225; i++) P[i] = 0.0;
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
        for (int i = 0; i < 225; i++) F[i] = 0.0;
        for (int i = 0; i < 15; i++) F[i * 15 + i] = 1.0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                F[i * 15 + j] -= wH.a[i * 3 + j] * dt;
                F[(3 + i) * 15 + j] = aH.a[i * 3 + j] * dt22;
                F[(6 + i) * 15 + j] = aH.a[i * 3 + j] * dt;
                F[i * 15 + 12 + j] = -invD.a[i * 3 + j] * dt;   // theta_H_biasOmega = -C.top
                F[(6 + i) * 15 + 9 + j] = -R.a[i * 3 + j] * dt; // vel_H_biasAcc = -B.bottom
            }
        for (int i = 0; i < 3; i++) F[(3 + i) * 15 + 6 + i] = dt;
        // bias Jacobians: H <- A H - [B | C]
        {
            double Hn[54];
            for (int i = 0; i < 9; i++)
                for (int j = 0; j < 6; j++) {
                    double a = 0.0;
                    for (int l = 0; l < 9; l++) a = fma(F[i * 15 + l], Hb[l * 6 + j], a);
                    Hn[i * 6 + j] = a;
                }
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    Hn[(3 + i) * 6 + j] -= R.a[i * 3 + j] * dt22;
                    Hn[(6 + i) * 6 + j] -= R.a[i * 3 + j] * dt;
                    Hn[i * 6 + 3 + j] -= invD.a[i * 3 + j] * dt;
                }
            for (int i = 0; i < 54; i++) Hb[i] = Hn[i];
        }
        // mean
        th = th + dt * wt;
        pos = pos + dt * vel + dt22 * anav;
        vel = vel + dt * anav;
        dtij += dt;
        // covariance: P <- F P F^T + G Q G^T
        for (int i = 0; i < 15; i++)
            for (int j = 0; j < 15; j++) {
                double a = 0.0;
                for (int l = 0; l < 15; l++) a = fma(F[i * 15 + l], P[l * 15 + j], a);
                T[i * 15 + j] = a;
            }
        for (int i = 0; i < 15; i++)
            for (int j = 0; j < 15; j++) {
                double a = 0.0;
                for (int l = 0; l < 15; l++) a = fma(T[i * 15 + l], F[j * 15 + l], a);
                P[i * 15 + j] = a;
            }
        const double sv = (prm.acc + prm.bias_int) * dt, sr = (prm.gyro + prm.bias_int) * dt;
        const M3 RRt = mulBT(R, R), DDt = mulBT(invD, invD);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                P[(6 + i) * 15 + 6 + j] += sv * RRt.a[i * 3 + j];   // (1/dt) vHb (aCov+int) vHb^T
                P[i * 15 + j] += sr * DDt.a[i * 3 + j];             // (1/dt) tHb (wCov+int) tHb^T
            }
            P[(3 + i) * 15 + 3 + i] += dt * prm.integration;
            P[(9 + i) * 15 + 9 + i] += dt * prm.bias_acc;
            P[(12 + i) * 15 + 12 + i] += dt * prm.bias_omega;
        }
    }
    // R upper with R^T R = P^-1: reverse Cholesky P = U U^T (U upper), R = U^-1
    bool ok = true;
    for (int i = 0; i < 225; i++) F[i] = 0.0;   // F <- U
    for (int j = 14; j >= 0; j--) {
        double d = P[j * 15 + j];
        for (int l = j + 1; l < 15; l++) d = fma(-F[j * 15 + l], F[j * 15 + l], d);
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        const double ujj = sqrt(d);
        F[j * 15 + j] = ujj;
        for (int i = 0; i < j; i++) {
            double a = 0.5 * (P[i * 15 + j] + P[j * 15 + i]);
            for (int l = j + 1; l < 15; l++) a = fma(-F[i * 15 + l], F[j * 15 + l], a);
            F[i * 15 + j] = a / ujj;
        }
    }
    for (int i = 0; i < 225; i++) T[i] = 0.0;   // T <- U^-1 (upper)
    for (int c = 0; c < 15; c++) {
        T[c * 15 + c] = 1.0 / F[c * 15 + c];
        for (int r = c - 1; r >= 0; r--) {
            double a = 0.0;
            for (int l = r + 1; l <= c; l++) a = fma(F[r * 15 + l], T[l * 15 + c], a);
            T[r * 15 + c] = -a / F[r * 15 + r];
        }
    }
    const long gk = g0 + f;
    double* out = v.imu_in + (size_t)(gk >> 6) * IMU_IN * TILE + (gk & 63);
#define OUTF(i) out[(size_t)(i) * TILE]
    OUTF(0) = dtij;
    OUTF(1) = th.x; OUTF(2) = th.y; OUTF(3) = th.z;
    OUTF(4) = pos.x; OUTF(5) = pos.y; OUTF(6) = pos.z;
    OUTF(7) = vel.x; OUTF(8) = vel.y; OUTF(9) = vel.z;
    for (int i = 0; i < 6; i++) OUTF(10 + i) = bh[i];
    for (int i = 0; i < 54; i++) OUTF(16 + i) = Hb[i];
    int o = 70;
    for (int r = 0; r < 15; r++)
        for (int c = r; c < 15; c++) OUTF(o++) = T[r * 15 + c];
#undef OUTF
    if (!ok || off[f + 1] == off[f]) atomicOr(status, 1);
}

// ------------------------------------------------------------------------------------ K1
// Algorithmic traffic per factor: 222 doubles in (2 states x 16, record 190), 465 out.
__global__ void __launch_bounds__(256) k_linearize_imu(View v, int which) {
    const long gk = (long)blockIdx.x * 256 + threadIdx.x;
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    if (k <= v.lo[w] || k >= v.hi[w]) return;
    const int b = v.sel[w] ^ which;

    const double* __restrict__ in = v.imu_in + (size_t)(gk >> 6) * IMU_IN * TILE + (gk & 63);
    double* __restrict__ out = v.imu_out + ((size_t)b * (size_t)(v.G >> 6) + (size_t)(gk >> 6)) * IMU_OUT * TILE + (gk & 63);
#define IN(f) in[(size_t)(f) * TILE]
#define OUT(f) out[(size_t)(f) * TILE]
#define JOUT(r, c) OUT(15 + (r) * 30 + (c))

    const State si = load_state(v, b, gk - 1), sj = load_state(v, b, gk);
    const double dt = IN(0);
    const V3 dba = si.ba - v3(IN(10), IN(11), IN(12));
    const V3 dbg = si.bg - v3(IN(13), IN(14), IN(15));
    // bias-corrected preintegrated delta: d + H (b_i - bhat)   (biasCorrectedDelta)
    double xt[9];
#pragma unroll
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
        for (int c = a; c < 9; c++) R[idx9(a, c)] = IN(70 + off15(a) + (c - a));

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
        for (int a = 0; a < 9; a++) JOUT(a, 3 + c) = o[a];
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
        for (int a = 0; a < 9; a++) JOUT(a, 12 + c) = (a <= 3 + c) ? -R[idx9(a < 3 + c ? a : 3 + c, 3 + c)] : 0.0;
        // V_j : rows v = -R_j^T
#pragma unroll
        for (int r = 0; r < 3; r++) { u[r] = 0; u[3 + r] = 0; u[6 + r] = -Rj.a[c * 3 + r]; }
        white9<6, 9>(R, u, o);
#pragma unroll
        for (int a = 0; a < 9; a++) JOUT(a, 15 + c) = o[a];
    }
#pragma unroll
    for (int c = 0; c < 18; c++)
#pragma unroll
        for (int a = 9; a < 15; a++) JOUT(a, c) = 0.0;

    // bias columns: B_i (18..23) and B_j (24..29)
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const int n_rc = 10 + c;  // rows 0..9+c of R(:, 9+c) are nonzero
        double Rc[15];
#pragma unroll
        for (int a = 0; a < 15; a++) Rc[a] = (a < n_rc) ? IN(70 + off15(a) + (9 + c - a)) : 0.0;
        const V3 hth = v3(IN(16 + 0 * 6 + c), IN(16 + 1 * 6 + c), IN(16 + 2 * 6 + c));
        const V3 hp = v3(IN(16 + 3 * 6 + c), IN(16 + 4 * 6 + c), IN(16 + 5 * 6 + c));
        const V3 hv = v3(IN(16 + 6 * 6 + c), IN(16 + 7 * 6 + c), IN(16 + 8 * 6 + c));
        const V3 u0 = mul(M5, hth), u1 = mul(Rji, hp), u2 = mul(Rji, hv);
        u[0] = u0.x; u[1] = u0.y; u[2] = u0.z; u[3] = u1.x; u[4] = u1.y; u[5] = u1.z; u[6] = u2.x; u[7] = u2.y; u[8] = u2.z;
        white9<0, 9>(R, u, o);
        const double rb = (c < 3) ? vget(rba, c) : vget(rbg, c - 3);
#pragma unroll
        for (int a = 0; a < 15; a++) {
            const double bi = (a < 9) ? o[a] + Rc[a] : Rc[a];
            JOUT(a, 18 + c) = bi;
            JOUT(a, 24 + c) = -Rc[a];
            rw[a] = fma(Rc[a], rb, rw[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 15; a++) OUT(a) = rw[a];
#undef IN
#undef OUT
#undef JOUT
}

// ------------------------------------------------------------------------------------ K2
// Algorithmic traffic per factor: 42 doubles in (2 poses x 7, record 28), 78 out.
__global__ void __launch_bounds__(256) k_linearize_between(View v, int which) {
    const long gk = (long)blockIdx.x * 256 + threadIdx.x;
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    const int lo = v.lo[w];
    if (k <= lo || k >= v.hi[w]) return;
    const int a = v.btw_a[gk];
    if (a < lo || a >= k) return;
    const int b = v.sel[w] ^ which;
    const long ga = (long)w * v.M + a;

    const double* __restrict__ in = v.btw_in + (size_t)(gk >> 6) * BTW_IN * TILE + (gk & 63);
    double* __restrict__ out = v.btw_out + ((size_t)b * (size_t)(v.G >> 6) + (size_t)(gk >> 6)) * BTW_OUT * TILE + (gk & 63);
#define IN(f) in[(size_t)(f) * TILE]
#define OUT(f) out[(size_t)(f) * TILE]

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
            OUT(6 + r * 6 + c) = sa;
            OUT(42 + r * 6 + c) = sb;
        }
    }
#undef IN
#undef OUT
}

// ------------------------------------------------------------------------------------ K2b
__global__ void k_linearize_prior(View v, int which) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= v.B) return;
    const int k = v.prior_k[w];
    if (k < v.lo[w] || k >= v.hi[w]) return;
    const int b = v.sel[w] ^ which;
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

// ------------------------------------------------------------------------------------ K3
// Owner-computes: lane = keyframe k. Deterministic (no atomics).
//   H[k][k]   = Jj^T Jj (imu k) + Ji^T Ji (imu k+1) + between (as b at k, as a at k+d) + prior
//   H[k][k-1] = Jj^T Ji (imu k) + Jb^T Ja (between a=k-1)
//   H[k][k-d] = Jb^T Ja (between a=k-d), pose 6x6 only
__host__ __device__ constexpr int imu_col(int side_j, int c) {
    return c < 9 ? c + (side_j ? 9 : 0) : c + 9 + (side_j ? 6 : 0);
}
__host__ __device__ constexpr int tri(int a, int b) { return a * (a + 1) / 2 + b; }  // a >= b

template <int SIDE_J>
VF_DI void acc_imu_diag(const double* __restrict__ f, double (&D)[120], double (&g)[15]) {
#pragma unroll 1
    for (int r = 0; r < 15; r++) {
        double x[15];
#pragma unroll
        for (int c = 0; c < 15; c++) x[c] = f[(size_t)(15 + r * 30 + imu_col(SIDE_J, c)) * TILE];
        const double rr = f[(size_t)r * TILE];
#pragma unroll
        for (int a = 0; a < 15; a++) {
            g[a] = fma(x[a], rr, g[a]);
#pragma unroll
            for (int b = 0; b <= a; b++) D[tri(a, b)] = fma(x[a], x[b], D[tri(a, b)]);
        }
    }
}

__global__ void __launch_bounds__(64) k_assemble(View v) {
    const long gk = (long)blockIdx.x * 64 + threadIdx.x;
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    const int lo = v.lo[w], hi = v.hi[w];
    if (k < lo || k >= hi) return;
    const int b = v.sel[w];
    const size_t tiles = (size_t)(v.G >> 6);
    const double* imu_out = v.imu_out + (size_t)b * tiles * IMU_OUT * TILE;
    const double* btw_out = v.btw_out + (size_t)b * tiles * BTW_OUT * TILE;
    auto imu_f = [&](long g) { return imu_out + (size_t)(g >> 6) * IMU_OUT * TILE + (g & 63); };
    auto btw_f = [&](long g) { return btw_out + (size_t)(g >> 6) * BTW_OUT * TILE + (g & 63); };
    double* Hk = v.H + (size_t)gk * HROW;

    double D[120], g[15];
#pragma unroll
    for (int i = 0; i < 120; i++) D[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 15; i++) g[i] = 0.0;

    const bool has_in = k > lo;         // imu factor k (k-1 -> k)
    const bool has_out = k + 1 < hi;    // imu factor k+1 (k -> k+1)
    if (has_in) acc_imu_diag<1>(imu_f(gk), D, g);
    if (has_out) acc_imu_diag<0>(imu_f(gk + 1), D, g);

    // between factor ending here (as b)
    const int a_here = has_in ? v.btw_a[gk] : -1;
    const bool btw_here = a_here >= lo && a_here < k;
    if (btw_here) {
        const double* f = btw_f(gk);
#pragma unroll 1
        for (int r = 0; r < 6; r++) {
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; c++) x[c] = f[(size_t)(42 + r * 6 + c) * TILE];
            const double rr = f[(size_t)r * TILE];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                g[a] = fma(x[a], rr, g[a]);
#pragma unroll
                for (int c = 0; c <= a; c++) D[tri(a, c)] = fma(x[a], x[c], D[tri(a, c)]);
            }
        }
    }
    // between factors starting here (as a), stored in the slot of their b = k + d
#pragma unroll 1
    for (int d = 1; d <= 3; d++) {
        if (k + d >= hi) break;
        if (v.btw_a[gk + d] != k) continue;
        const double* f = btw_f(gk + d);
#pragma unroll 1
        for (int r = 0; r < 6; r++) {
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; c++) x[c] = f[(size_t)(6 + r * 6 + c) * TILE];
            const double rr = f[(size_t)r * TILE];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                g[a] = fma(x[a], rr, g[a]);
#pragma unroll
                for (int c = 0; c <= a; c++) D[tri(a, c)] = fma(x[a], x[c], D[tri(a, c)]);
            }
        }
    }
    if (v.prior_k[w] == k) {
        const double* f = v.prior_out + ((size_t)b * v.B + w) * PRIOR_OUT;
#pragma unroll 1
        for (int r = 0; r < 15; r++) {
            double x[15];
#pragma unroll
            for (int c = 0; c < 15; c++) x[c] = f[15 + r * 15 + c];
            const double rr = f[r];
#pragma unroll
            for (int a = 0; a < 15; a++) {
                g[a] = fma(x[a], rr, g[a]);
#pragma unroll
                for (int c = 0; c <= a; c++) D[tri(a, c)] = fma(x[a], x[c], D[tri(a, c)]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 15; a++) {
        v.gvec[(size_t)gk * 15 + a] = g[a];
#pragma unroll
        for (int c = 0; c < 15; c++) Hk[a * 15 + c] = (c <= a) ? D[tri(a, c)] : D[tri(c, a)];
    }

    // off-diagonal block d=1: Jj^T Ji of imu factor k (+ between a = k-1 in the pose 6x6)
    if (has_in) {
        const double* f = imu_f(gk);
#pragma unroll 1
        for (int cb = 0; cb < 3; cb++) {  // 5 columns of Ji at a time
            double O[75];
#pragma unroll
            for (int i = 0; i < 75; i++) O[i] = 0.0;
#pragma unroll 1
            for (int r = 0; r < 15; r++) {
                double xj[15], xi[5];
#pragma unroll
                for (int c = 0; c < 15; c++) xj[c] = f[(size_t)(15 + r * 30 + imu_col(1, c)) * TILE];
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    const int cc = cb * 5 + c;  // runtime (cb) -> compute the column index arithmetically
                    const int col = cc < 9 ? cc : cc + 9;
                    xi[c] = f[(size_t)(15 + r * 30 + col) * TILE];
                }
#pragma unroll
                for (int a = 0; a < 15; a++)
#pragma unroll
                    for (int c = 0; c < 5; c++) O[a * 5 + c] = fma(xj[a], xi[c], O[a * 5 + c]);
            }
            if (btw_here && a_here == k - 1 && cb < 2) {
                const double* fb = btw_f(gk);
#pragma unroll 1
                for (int r = 0; r < 6; r++) {
                    double xb[6];
#pragma unroll
                    for (int c = 0; c < 6; c++) xb[c] = fb[(size_t)(42 + r * 6 + c) * TILE];
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        const int cc = cb * 5 + c;
                        if (cc < 6) {
                            const double xa = fb[(size_t)(6 + r * 6 + cc) * TILE];
#pragma unroll
                            for (int a = 0; a < 6; a++) O[a * 5 + c] = fma(xb[a], xa, O[a * 5 + c]);
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < 15; a++)
#pragma unroll
                for (int c = 0; c < 5; c++) Hk[225 + a * 15 + cb * 5 + c] = O[a * 5 + c];
        }
    }
    // blocks d=2,3: pose 6x6 from a between factor with a = k-d (zero otherwise)
#pragma unroll 1
    for (int d = 2; d <= 3; d++) {
        double O[36];
#pragma unroll
        for (int i = 0; i < 36; i++) O[i] = 0.0;
        if (btw_here && a_here == k - d) {
            const double* fb = btw_f(gk);
#pragma unroll 1
            for (int r = 0; r < 6; r++) {
                double xb[6], xa[6];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    xb[c] = fb[(size_t)(42 + r * 6 + c) * TILE];
                    xa[c] = fb[(size_t)(6 + r * 6 + c) * TILE];
                }
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int c = 0; c < 6; c++) O[a * 6 + c] = fma(xb[a], xa[c], O[a * 6 + c]);
            }
        }
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 6; c++) Hk[d * 225 + a * 15 + c] = O[a * 6 + c];
    }
}

// ------------------------------------------------------------------------------------ K4
VF_DI double readlane_d(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// v_rsq_f64 is good to 2^29 ulp (rel. 2^-23); two Newton steps reach full double precision.
VF_DI double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double t = x * y;
    double e = fma(-t, y, 1.0);
    y = fma(0.5 * y, e, y);
    t = x * y;
    e = fma(-t, y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}

// One wavefront per window.  Right-looking block Cholesky of the block-banded normal matrix,
// exploiting its profile: IMU factors couple consecutive keyframes in all 15 dof, between
// factors couple keyframes up to 3 apart in the 6 pose dof only, so the active set while
// eliminating keyframe k is  [k: 15] [k+1: 15] [k+2: pose 6] [k+3: pose 6]  = 42 rows, plus
// the right-hand side carried as a 43rd row (forward substitution for free).
//   * the 43x15 panel is factorised in registers: lane p owns row p, pivots/multipliers are
//     broadcast with v_readlane, 1/sqrt by v_rsq_f64 + 2 Newton steps (no IEEE sqrt/div chain);
//   * the 405 trailing entries are spread evenly over the 64 lanes (7 dot products each) and
//     applied to a circular 4-keyframe window kept in LDS;
//   * the next block row of H is prefetched from HBM before the panel and dropped into the
//     slot the pivot keyframe frees.
// Sequential in k, hence latency-bound; see DESIGN.md "K4" for the cycle budget.
constexpr int PROWS = 43;
constexpr int LDW = 61;

__global__ void __launch_bounds__(64) k_band_solve(View v, int ablate) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const int lo = v.lo[w], hi = v.hi[w], n = hi - lo;
    if (n <= 0) return;
    __shared__ double Wd[60 * LDW];
    __shared__ double Gd[64];
    __shared__ double P[PROWS * 15 + 3];
    __shared__ double dl[64];
    const double lam = v.lambda[w];
    const size_t base = (size_t)w * v.M + lo;
    int failed = 0;

    // panel row of this lane: segment (keyframe offset) and dof
    const int pd = lane < 15 ? 0 : (lane < 30 ? 1 : (lane < 36 ? 2 : 3));
    const int pa = lane < 15 ? lane : (lane < 30 ? lane - 15 : (lane < 36 ? lane - 30 : lane - 36));
    // the 7 trailing entries of this lane: e = lane + 64 j -> (row pr in 15..42, col pc in 15..min(pr,41))
    int t_rd[7], t_ra[7], t_cd[7], t_ca[7], t_pr[7], t_pc[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        int e = lane + 64 * j, r = 0;
        if (e >= 405) e = 404;                       // duplicates are masked below
        while (r < 26 && e > r) { e -= r + 1; r++; } // row r of the 27-triangle, then the rhs row
        int pr, pc;
        if (r < 26 || e <= 26) { pr = 15 + r; pc = 15 + e; }
        if (r == 26 && e > 26) { pr = 42; pc = 15 + (e - 27); }
        t_pr[j] = pr; t_pc[j] = pc;
        t_rd[j] = pr < 30 ? 1 : (pr < 36 ? 2 : 3);
        t_ra[j] = pr < 30 ? pr - 15 : (pr < 36 ? pr - 30 : pr - 36);
        t_cd[j] = pc < 30 ? 1 : (pc < 36 ? 2 : 3);
        t_ca[j] = pc < 30 ? pc - 15 : (pc < 36 ? pc - 30 : pc - 36);
    }
    // block-row load map: idx = lane + 64 j over a 15x15 block
    int l_a[4], l_c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int idx = lane + 64 * j;
        l_a[j] = idx / 15;
        l_c[j] = idx - l_a[j] * 15;
    }
    const int q_a = lane / 6, q_c = lane - q_a * 6;  // 6x6 pose block map (lane < 36)

    double h0[4], h1[4], h2 = 0.0, h3 = 0.0, hg = 0.0;
    auto fetch_row = [&](int kk) {   // HBM -> registers
        if (kk < n) {
            const double* Hk = v.H + (base + kk) * HROW;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int idx = lane + 64 * j;
                const bool in = idx < 225;
                h0[j] = in ? Hk[idx] : 0.0;
                h1[j] = (in && kk >= 1) ? Hk[225 + idx] : 0.0;
            }
            h2 = (lane < 36 && kk >= 2) ? Hk[450 + q_a * 15 + q_c] : 0.0;
            h3 = (lane < 36 && kk >= 3) ? Hk[675 + q_a * 15 + q_c] : 0.0;
            hg = lane < 15 ? -v.gvec[(base + kk) * 15 + lane] : 0.0;
        } else {   // beyond the window: identity rows
#pragma unroll
            for (int j = 0; j < 4; j++) { h0[j] = 0.0; h1[j] = 0.0; }
            h2 = h3 = hg = 0.0;
        }
    };
    auto commit_row = [&](int kk) {  // registers -> LDS slot of keyframe kk
        const int s = (kk & 3) * 15;
        const int c1 = ((kk - 1) & 3) * 15, c2 = ((kk - 2) & 3) * 15, c3 = ((kk - 3) & 3) * 15;
        const bool real = kk < n;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (lane + 64 * j < 225) {
                double d0 = h0[j];
                if (l_a[j] == l_c[j]) d0 = real ? d0 + lam : 1.0;
                Wd[(s + l_a[j]) * LDW + s + l_c[j]] = d0;
                Wd[(s + l_a[j]) * LDW + c1 + l_c[j]] = h1[j];
            }
        }
        // d = 2, 3: pose rows only; columns 6..14 start at zero and fill in during elimination
#pragma unroll
        for (int it = 0; it < 2; it++) {   // 6x15 entries; every lane takes part in the shuffles
            const int e = lane + 64 * it;
            const int a = e / 15, c = e - a * 15;
            const bool valid = e < 90, ld = valid && c < 6;
            const int owner = ld ? a * 6 + c : 0;   // lanes < 36 hold the 6x6 pose block (q_a, q_c)
            const double x2 = __shfl(h2, owner), x3 = __shfl(h3, owner);
            if (valid) {
                Wd[(s + a) * LDW + c2 + c] = ld ? x2 : 0.0;
                Wd[(s + a) * LDW + c3 + c] = ld ? x3 : 0.0;
            }
        }
        if (lane < 15) Gd[s + lane] = hg;
    };
    for (int kk = 0; kk < 4; kk++) {
        fetch_row(kk);
        commit_row(kk);
    }
    __syncthreads();

    for (int k = 0; k < n; k++) {
        const int s0 = (k & 3) * 15;
        fetch_row(k + 4);  // in flight during the panel factorisation
        const int ri = ((((k + pd) & 3) * 15) + pa) * LDW;
        double p[15];
#pragma unroll
        for (int c = 0; c < 15; c++)
            p[c] = lane < 42 ? Wd[ri + s0 + c] : (lane == 42 ? Gd[s0 + c] : 0.0);
        double dinv = 0.0;
        if (!(ablate & 4))
#pragma unroll
        for (int c = 0; c < 15; c++) {
            double dv = readlane_d(p[c], c);
            if (!(dv > 0.0)) { failed = 1; dv = 1.0; }
            const double inv = fast_rsqrt(dv);
            p[c] *= inv;
            if (lane == c) dinv = inv;
#pragma unroll
            for (int c2 = c + 1; c2 < 15; c2++) {
                const double l = readlane_d(p[c], c2);
                p[c2] = fma(-p[c], l, p[c2]);
            }
        }
        if (lane < PROWS) {
#pragma unroll
            for (int c = 0; c < 15; c++) P[lane * 15 + c] = (lane == c) ? dinv : p[c];  // diagonal holds 1/L_cc
        }
        __syncthreads();
        if (!(ablate & 8)) {   // panel -> HBM (coalesced), kept for the back substitution
            double* Lk = v.Lp + (base + k) * PANEL;
            for (int e = lane; e < PANEL; e += 64) Lk[e] = P[e];
        }
        if (!(ablate & 1))
#pragma unroll
        for (int j = 0; j < 7; j++) {
            if (lane + 64 * j < 405) {
                const double* a = P + t_pr[j] * 15;
                const double* b = P + t_pc[j] * 15;
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int c = 0; c < 14; c += 2) {
                    s1 = fma(a[c], b[c], s1);
                    s2 = fma(a[c + 1], b[c + 1], s2);
                }
                s1 = fma(a[14], b[14], s1) + s2;
                const int cj = (((k + t_cd[j]) & 3) * 15) + t_ca[j];
                if (t_pr[j] < 42) Wd[((((k + t_rd[j]) & 3) * 15) + t_ra[j]) * LDW + cj] -= s1;
                else Gd[cj] -= s1;
            }
        }
        __syncthreads();
        commit_row(k + 4);  // the pivot keyframe's slot is free now
        __syncthreads();
    }

    // back substitution: delta_k = L_kk^-T (y_k - sum_p L[p][k-cols]^T delta(p)), p over rows 15..41
    if (ablate & 2) { if (lane == 0) v.fail[w] = failed; return; }
    if (lane < 60) dl[lane] = 0.0;
    const int part = lane >> 4, cc = lane & 15;          // 4 partial sums per column
    const int p_lo = 15 + part * 7, p_hi = part == 3 ? 42 : p_lo + 7;
    double nxt[11];
    {
        const double* Lk = v.Lp + (base + n - 1) * PANEL;
#pragma unroll
        for (int j = 0; j < 11; j++) nxt[j] = (lane + 64 * j < PANEL) ? Lk[lane + 64 * j] : 0.0;
    }
    __syncthreads();
    for (int k = n - 1; k >= 0; k--) {
#pragma unroll
        for (int j = 0; j < 11; j++)
            if (lane + 64 * j < PANEL) P[lane + 64 * j] = nxt[j];
        __syncthreads();
        if (k > 0) {  // prefetch the next panel while this one is consumed
            const double* Lk = v.Lp + (base + k - 1) * PANEL;
#pragma unroll
            for (int j = 0; j < 11; j++) nxt[j] = (lane + 64 * j < PANEL) ? Lk[lane + 64 * j] : 0.0;
        }
        double s = 0.0;
        if (cc < 15) {
            for (int pp = p_lo; pp < p_hi; pp++) {
                const int d = pp < 30 ? 1 : (pp < 36 ? 2 : 3);
                const int a = pp < 30 ? pp - 15 : (pp < 36 ? pp - 30 : pp - 36);
                s = fma(P[pp * 15 + cc], dl[((k + d) & 3) * 15 + a], s);
            }
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        double lcol[15];   // column `lane` of L_kk below the diagonal, and 1/L_cc on the diagonal
#pragma unroll
        for (int c = 0; c < 15; c++) lcol[c] = lane < 15 ? P[c * 15 + lane] : 0.0;
        s = lane < 15 ? P[42 * 15 + lane] - s : 0.0;
#pragma unroll
        for (int c = 14; c >= 0; c--) {
            const double xc = readlane_d(s, c) * readlane_d(lcol[c], c);
            if (lane == c) s = xc;
            else if (lane < c) s = fma(-lcol[c], xc, s);
        }
        __syncthreads();
        if (lane < 15) {
            dl[(k & 3) * 15 + lane] = s;
            v.delta[(base + k) * 15 + lane] = s;
        }
        __syncthreads();
    }
    if (lane == 0) v.fail[w] = failed;
}

// ------------------------------------------------------------------------------------ K5
__global__ void __launch_bounds__(256) k_retract(View v) {
    const long gk = (long)blockIdx.x * 256 + threadIdx.x;
    if (gk >= v.G) return;
    const int w = (int)(gk / v.M), k = (int)(gk - (long)w * v.M);
    if (k < v.lo[w] || k >= v.hi[w]) return;
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

// cost of buffer (sel ^ !init) per window, then the LM decision. One 256-thread block per window;
// fixed-shape tree reduction => bitwise reproducible.
__global__ void __launch_bounds__(256) k_decide(View v, int init) {
    const int w = blockIdx.x, tid = threadIdx.x;
    const int lo = v.lo[w], hi = v.hi[w];
    const int b = init ? v.sel[w] : (v.sel[w] ^ 1);
    const size_t tiles = (size_t)(v.G >> 6);
    const double* imu_out = v.imu_out + (size_t)b * tiles * IMU_OUT * TILE;
    const double* btw_out = v.btw_out + (size_t)b * tiles * BTW_OUT * TILE;
    double s = 0.0;
    for (int k = lo + 1 + tid; k < hi; k += 256) {
        const long gk = (long)w * v.M + k;
        const double* f = imu_out + (size_t)(gk >> 6) * IMU_OUT * TILE + (gk & 63);
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
    }
    __shared__ double red[256];
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) {
        const double c = 0.5 * red[0];
        if (init) {
            v.cost[w] = c;
            v.fail[w] = 0;
        } else {
            const bool ok = (v.fail[w] == 0) && (c < v.cost[w]);
            if (v.fail[w]) v.n_fail[w] += 1;
            if (ok) {
                v.sel[w] ^= 1;
                v.cost[w] = c;
                v.n_acc[w] += 1;
                const double l = v.lambda[w] / v.lam_down;
                v.lambda[w] = l < v.lam_min ? v.lam_min : l;
            } else {
                v.n_rej[w] += 1;
                const double l = v.lambda[w] * v.lam_up;
                v.lambda[w] = l > v.lam_max ? v.lam_max : l;
            }
            v.fail[w] = 0;
        }
    }
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
__global__ void k_predict(View v, int window, int k0, int n) {
    const int w = window >= 0 ? window : (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w >= v.B || (window >= 0 && (blockIdx.x | threadIdx.x))) return;
    const int b = v.sel[w];
    for (int k = k0; k < k0 + n; k++) {
        const long gk = (long)w * v.M + k;
        const State si = load_state(v, b, gk - 1);
        store_state(v, b, gk, predict_state(v, si, gk));
    }
}

// fixed-lag slide by one keyframe: hi += 1 (predict the new state), lo += 1, re-anchor the prior
__global__ void k_slide(View v, const double* sigma15) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= v.B) return;
    const int b = v.sel[w];
    const int hi = v.hi[w], lo = v.lo[w];
    if (hi >= v.M) return;  // out of slots: caller must compact
    const long gnew = (long)w * v.M + hi;
    store_state(v, b, gnew, predict_state(v, load_state(v, b, gnew - 1), gnew));
    v.hi[w] = hi + 1;
    v.lo[w] = lo + 1;
    const long ganchor = (long)w * v.M + lo + 1;
    v.prior_k[w] = lo + 1;
    double* pin = v.prior_in + (size_t)w * PRIOR_IN;
    for (int c = 0; c < 16; c++) pin[c] = XS(b, c, ganchor);
    for (int c = 0; c < 15; c++) pin[16 + c] = sigma15[c];
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
    if (n > 0) hipLaunchKernelGGL(k_preintegrate, dim3(nblk(n, 64)), dim3(64), 0, s, v, g0, n, off, steps, bhat6, prm, status);
}
void launch_linearize_imu(const View& v, int which, hipStream_t s) {
    hipLaunchKernelGGL(k_linearize_imu, dim3(nblk(v.G, 256)), dim3(256), 0, s, v, which);
}
void launch_linearize_between(const View& v, int which, hipStream_t s) {
    hipLaunchKernelGGL(k_linearize_between, dim3(nblk(v.G, 256)), dim3(256), 0, s, v, which);
}
void launch_linearize_prior(const View& v, int which, hipStream_t s) {
    hipLaunchKernelGGL(k_linearize_prior, dim3(nblk(v.B, 64)), dim3(64), 0, s, v, which);
}
void launch_assemble(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_assemble, dim3(nblk(v.G, 64)), dim3(64), 0, s, v);
}
void launch_band_solve(const View& v, hipStream_t s) {
    static const int ablate = getenv("VF_SOLVE_ABLATE") ? atoi(getenv("VF_SOLVE_ABLATE")) : 0;  // timing experiments only
    hipLaunchKernelGGL(k_band_solve, dim3(v.B), dim3(64), 0, s, v, ablate);
}
void launch_retract(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(k_retract, dim3(nblk(v.G, 256)), dim3(256), 0, s, v);
}
void launch_decide(const View& v, int init, hipStream_t s) {
    hipLaunchKernelGGL(k_decide, dim3(v.B), dim3(256), 0, s, v, init);
}
void launch_predict(const View& v, int window, int k0, int n, hipStream_t s) {
    if (window >= 0) hipLaunchKernelGGL(k_predict, dim3(1), dim3(1), 0, s, v, window, k0, n);
    else hipLaunchKernelGGL(k_predict, dim3(nblk(v.B, 64)), dim3(64), 0, s, v, window, k0, n);
}
void launch_slide(const View& v, const double* sigma15_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_slide, dim3(nblk(v.B, 64)), dim3(64), 0, s, v, sigma15_dev);
}
void launch_scatter(const double* aos, double* aosoa, long g0, long n, int nf, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_scatter, dim3(nblk(n * nf, 256)), dim3(256), 0, s, aos, aosoa, g0, n, nf);
}
void launch_gather(const double* aosoa, double* aos, long g0, long n, int nf, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather, dim3(nblk(n * nf, 256)), dim3(256), 0, s, aosoa, aos, g0, n, nf);
}
void launch_scatter_states(const double* aos, double* x, long G, int buf, long g0, long n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_scatter_states, dim3(nblk(n * 16, 256)), dim3(256), 0, s, aos, x, G, buf, g0, n);
}
void launch_gather_states(const double* x, double* aos, long G, const int* sel, int M, int which, long g0, long n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather_states, dim3(nblk(n * 16, 256)), dim3(256), 0, s, x, aos, G, sel, M, which, g0, n);
}

}  // namespace vf
