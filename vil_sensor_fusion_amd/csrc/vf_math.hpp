// vf_math.hpp -- device-side SO(3)/SE(3) algebra for the gfx950 kernels (float64).
//
// Everything is written on unit quaternions and closed-form 3x3 blocks so that a factor
// lives entirely in one lane's registers (no local arrays that could spill to scratch):
// all loops are fully unrolled over compile-time bounds.  The formulas are the ones GTSAM
// evaluates behind the reference's call sites (IMUManager.cpp:68-73, GraphManager.cpp:86,
// GraphManager.cpp:33-35); the CPU oracle follows GTSAM's function structure instead, so
// the two implementations are independent.
#pragma once
#include <hip/hip_runtime.h>

namespace vf {

#define VF_DI __device__ __forceinline__

struct V3 { double x, y, z; };
struct M3 { double a[9]; };  // row-major
struct Q4 { double w, x, y, z; };

VF_DI V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
VF_DI V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
VF_DI V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
VF_DI V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
VF_DI V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
VF_DI double dot(V3 a, V3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
VF_DI V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
VF_DI double vget(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

VF_DI V3 mul(const M3& A, V3 v) {
    return v3(fma(A.a[0], v.x, fma(A.a[1], v.y, A.a[2] * v.z)),
              fma(A.a[3], v.x, fma(A.a[4], v.y, A.a[5] * v.z)),
              fma(A.a[6], v.x, fma(A.a[7], v.y, A.a[8] * v.z)));
}
VF_DI V3 mulT(const M3& A, V3 v) {  // A^T v
    return v3(fma(A.a[0], v.x, fma(A.a[3], v.y, A.a[6] * v.z)),
              fma(A.a[1], v.x, fma(A.a[4], v.y, A.a[7] * v.z)),
              fma(A.a[2], v.x, fma(A.a[5], v.y, A.a[8] * v.z)));
}
VF_DI M3 mul(const M3& A, const M3& B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C.a[i * 3 + j] = fma(A.a[i * 3], B.a[j], fma(A.a[i * 3 + 1], B.a[3 + j], A.a[i * 3 + 2] * B.a[6 + j]));
    return C;
}
VF_DI M3 mulTA(const M3& A, const M3& B) {  // A^T B
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C.a[i * 3 + j] = fma(A.a[i], B.a[j], fma(A.a[3 + i], B.a[3 + j], A.a[6 + i] * B.a[6 + j]));
    return C;
}
VF_DI M3 mulBT(const M3& A, const M3& B) {  // A B^T
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C.a[i * 3 + j] = fma(A.a[i * 3], B.a[j * 3], fma(A.a[i * 3 + 1], B.a[j * 3 + 1], A.a[i * 3 + 2] * B.a[j * 3 + 2]));
    return C;
}
VF_DI M3 transpose(const M3& A) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C.a[i * 3 + j] = A.a[j * 3 + i];
    return C;
}
VF_DI M3 skew(V3 v) {
    M3 S;
    S.a[0] = 0;    S.a[1] = -v.z; S.a[2] = v.y;
    S.a[3] = v.z;  S.a[4] = 0;    S.a[5] = -v.x;
    S.a[6] = -v.y; S.a[7] = v.x;  S.a[8] = 0;
    return S;
}
// A * skew(v): column j of the product is A (e_? x ...) ; written out to avoid zero FMAs
VF_DI M3 mulSkew(const M3& A, V3 v) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double a0 = A.a[i * 3], a1 = A.a[i * 3 + 1], a2 = A.a[i * 3 + 2];
        C.a[i * 3 + 0] = a1 * v.z - a2 * v.y;
        C.a[i * 3 + 1] = a2 * v.x - a0 * v.z;
        C.a[i * 3 + 2] = a0 * v.y - a1 * v.x;
    }
    return C;
}

// ---- series coefficients in x = theta^2 (Horner below SERIES_X, closed form above) ----
constexpr double SERIES_X = 0.25;

struct CoefABC { double A, B, C; };
VF_DI CoefABC coef_abc(double x) {
    CoefABC c;
    if (x < SERIES_X) {
        c.A = fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, 1.0 / 6227020800.0, -1.0 / 39916800.0), 1.0 / 362880.0), -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
        c.B = fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, 1.0 / 87178291200.0, -1.0 / 479001600.0), 1.0 / 3628800.0), -1.0 / 40320.0), 1.0 / 720.0), -1.0 / 24.0), 0.5);
        c.C = fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, 1.0 / 1307674368000.0, -1.0 / 6227020800.0), 1.0 / 39916800.0), -1.0 / 362880.0), 1.0 / 5040.0), -1.0 / 120.0), 1.0 / 6.0);
    } else {
        const double th = sqrt(x);
        double s, co;
        sincos(th, &s, &co);
        c.A = s / th;
        c.B = (1.0 - co) / x;
        c.C = (1.0 - c.A) / x;
    }
    return c;
}
// dB = B'(th)/th, dC = C'(th)/th
struct CoefD { double dB, dC; };
VF_DI CoefD coef_dbdc(double x, const CoefABC& c) {
    CoefD d;
    if (x < SERIES_X) {
        d.dB = fma(x, fma(x, fma(x, fma(x, fma(x, 1.0 / 7264857600.0, -1.0 / 47900160.0), 1.0 / 453600.0), -1.0 / 6720.0), 1.0 / 180.0), -1.0 / 12.0);
        d.dC = fma(x, fma(x, fma(x, fma(x, fma(x, 1.0 / 108972864000.0, -1.0 / 622702080.0), 1.0 / 4989600.0), -1.0 / 60480.0), 1.0 / 1260.0), -1.0 / 60.0);
    } else {
        d.dB = (c.A - 2.0 * c.B) / x;
        d.dC = (c.B - 3.0 * c.C) / x;
    }
    return d;
}
// E = 1/th^2 - (1+cos)/(2 th sin): coefficient of W^2 in J_r^{-1}; dE = E'(th)/th
VF_DI double coef_e(double x) {
    if (x < SERIES_X)
        return fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, 3617.0 / 10670622842880000.0, 1.0 / 74724249600.0), 691.0 / 1307674368000.0), 1.0 / 47900160.0), 1.0 / 1209600.0), 1.0 / 30240.0), 1.0 / 720.0), 1.0 / 12.0);
    const double th = sqrt(x);
    double s, co;
    sincos(th, &s, &co);
    return 1.0 / x - (1.0 + co) / (2.0 * th * s);
}
VF_DI double coef_de(double x) {
    if (x < SERIES_X)
        return fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, 50638.0 / 10670622842880000.0, 1.0 / 6227020800.0), 691.0 / 130767436800.0), 1.0 / 5987520.0), 1.0 / 201600.0), 1.0 / 7560.0), 1.0 / 360.0);
    const double th = sqrt(x);
    double s, co;
    sincos(th, &s, &co);
    const double A = s / th, B = (1.0 - co) / x;
    return ((1.0 + A) / (2.0 * B) - 2.0) / (x * x);
}

// ---- quaternions ----
VF_DI Q4 q4(double w, double x, double y, double z) { Q4 q; q.w = w; q.x = x; q.y = y; q.z = z; return q; }
VF_DI Q4 qmul(Q4 a, Q4 b) {
    return q4(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
              a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w);
}
VF_DI Q4 qconj(Q4 a) { return q4(a.w, -a.x, -a.y, -a.z); }
VF_DI Q4 qnormalize(Q4 a) {
    const double n = 1.0 / sqrt(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z);
    return q4(a.w * n, a.x * n, a.y * n, a.z * n);
}
VF_DI M3 qrot(Q4 q) {  // rotation matrix of a (not necessarily unit) quaternion
    const double n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z, s = 2.0 / n;
    M3 R;
    R.a[0] = 1 - s * (q.y * q.y + q.z * q.z); R.a[1] = s * (q.x * q.y - q.w * q.z);     R.a[2] = s * (q.x * q.z + q.w * q.y);
    R.a[3] = s * (q.x * q.y + q.w * q.z);     R.a[4] = 1 - s * (q.x * q.x + q.z * q.z); R.a[5] = s * (q.y * q.z - q.w * q.x);
    R.a[6] = s * (q.x * q.z - q.w * q.y);     R.a[7] = s * (q.y * q.z + q.w * q.x);     R.a[8] = 1 - s * (q.x * q.x + q.y * q.y);
    return R;
}
// Exp: rotation vector -> unit quaternion (cos(th/2), sin(th/2)/th * w)
VF_DI Q4 qexp(V3 w) {
    const double x = dot(w, w);
    double c, k;
    if (x < 1e-4) {  // series in x/4
        const double y = 0.25 * x;
        c = fma(y, fma(y, fma(y, -1.0 / 720.0, 1.0 / 24.0), -0.5), 1.0);
        k = 0.5 * fma(y, fma(y, fma(y, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
    } else {
        const double th = sqrt(x);
        double s;
        sincos(0.5 * th, &s, &c);
        k = s / th;
    }
    return q4(c, k * w.x, k * w.y, k * w.z);
}
// Log: unit quaternion -> rotation vector in (-pi, pi]
VF_DI V3 qlog(Q4 q) {
    if (q.w < 0) q = q4(-q.w, -q.x, -q.y, -q.z);
    const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    double f;
    if (n < 1e-7) {
        const double r = n / q.w;
        f = 2.0 / q.w * (1.0 - r * r / 3.0);
    } else {
        f = 2.0 * atan2(n, q.w) / n;
    }
    return v3(f * q.x, f * q.y, f * q.z);
}

// ---- SO(3) Jacobians ----
VF_DI M3 so3_jr(V3 w) {  // I - B W + C W^2
    const double x = dot(w, w);
    const CoefABC c = coef_abc(x);
    M3 J;
    const double xx = w.x * w.x, yy = w.y * w.y, zz = w.z * w.z;
    const double xy = w.x * w.y, xz = w.x * w.z, yz = w.y * w.z;
    J.a[0] = 1 - c.C * (yy + zz);         J.a[1] = c.B * w.z + c.C * xy;        J.a[2] = -c.B * w.y + c.C * xz;
    J.a[3] = -c.B * w.z + c.C * xy;       J.a[4] = 1 - c.C * (xx + zz);         J.a[5] = c.B * w.x + c.C * yz;
    J.a[6] = c.B * w.y + c.C * xz;        J.a[7] = -c.B * w.x + c.C * yz;       J.a[8] = 1 - c.C * (xx + yy);
    return J;
}
VF_DI M3 so3_jr_inv_e(V3 w, double E) {  // I + W/2 + E W^2 with a given E
    M3 J;
    const double xx = w.x * w.x, yy = w.y * w.y, zz = w.z * w.z;
    const double xy = w.x * w.y, xz = w.x * w.z, yz = w.y * w.z;
    J.a[0] = 1 - E * (yy + zz);           J.a[1] = -0.5 * w.z + E * xy;         J.a[2] = 0.5 * w.y + E * xz;
    J.a[3] = 0.5 * w.z + E * xy;          J.a[4] = 1 - E * (xx + zz);           J.a[5] = -0.5 * w.x + E * yz;
    J.a[6] = -0.5 * w.y + E * xz;         J.a[7] = 0.5 * w.x + E * yz;          J.a[8] = 1 - E * (xx + yy);
    return J;
}
VF_DI M3 so3_jr_inv(V3 w) { return so3_jr_inv_e(w, coef_e(dot(w, w))); }

// d/dtheta [ J_r(theta) c ] for fixed c (so3::DexpFunctor::applyDexp H1), used by the
// tangent preintegration's A matrix
VF_DI M3 so3_jr_apply_dtheta(V3 th, V3 c) {
    const double x = dot(th, th);
    const CoefABC k = coef_abc(x);
    const CoefD d = coef_dbdc(x, k);
    const V3 txc = cross(th, c), ttxc = cross(th, txc);
    const double tc = dot(th, c);
    M3 D;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double v = -vget(txc, i) * d.dB * vget(th, j) + vget(ttxc, i) * d.dC * vget(th, j) +
                       k.C * (vget(th, i) * vget(c, j) - 2.0 * vget(c, i) * vget(th, j));
            if (i == j) v += k.C * tc;
            D.a[i * 3 + j] = v;
        }
    const M3 S = skew(c);
#pragma unroll
    for (int i = 0; i < 9; i++) D.a[i] += k.B * S.a[i];
    return D;
}

// ---- SE(3) ----
// Logmap of (q, t): xi = [w, u], u = (I - W/2 + E W^2) t
struct Xi6 { V3 w, u; };
VF_DI Xi6 se3_log(Q4 q, V3 t) {
    Xi6 r;
    r.w = qlog(q);
    const double E = coef_e(dot(r.w, r.w));
    const V3 wt = cross(r.w, t);
    const V3 wwt = cross(r.w, wt);
    r.u = t - 0.5 * wt + E * wwt;
    return r;
}
// Expmap: (Exp(w), J_l(w) v) with J_l = I + B W + C W^2
VF_DI void se3_exp(V3 w, V3 v, Q4* q, V3* t) {
    *q = qexp(w);
    const CoefABC c = coef_abc(dot(w, w));
    const V3 wv = cross(w, v);
    const V3 wwv = cross(w, wv);
    *t = v + c.B * wv + c.C * wwv;
}
// Pose3::LogmapDerivative = [[Jw, 0], [Q2, Jw]].  Q2 is the directional derivative of
// J_r^{-1}(w) along u (the adjoint algebra is the dual-number extension W + eps U):
//   Q2 = U/2 + E (W U + U W) + dE (w.u) W^2
VF_DI void se3_jr_inv(Xi6 xi, M3* Jw, M3* Q2) {
    const double x = dot(xi.w, xi.w);
    const double E = coef_e(x), dE = coef_de(x);
    *Jw = so3_jr_inv_e(xi.w, E);
    const V3 w = xi.w, u = xi.u;
    const double wu = dot(w, u), k = dE * wu;
    // W U + U W = w u^T + u w^T - 2 (w.u) I ; W^2 = w w^T - x I
    M3 Q;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double wi = vget(w, i), wj = vget(w, j), ui = vget(u, i), uj = vget(u, j);
            double v = E * (wi * uj + ui * wj) + k * (wi * wj);
            if (i == j) v += -2.0 * E * wu - k * x;
            Q.a[i * 3 + j] = v;
        }
    Q.a[1] += -0.5 * u.z; Q.a[2] += 0.5 * u.y;
    Q.a[3] += 0.5 * u.z;  Q.a[5] += -0.5 * u.x;
    Q.a[6] += -0.5 * u.y; Q.a[7] += 0.5 * u.x;
    *Q2 = Q;
}

}  // namespace vf
