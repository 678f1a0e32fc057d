/*
 * vf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see vf_oracle.h).
 *
 * Restates, in plain C / float64, what the reference's hot path asks GTSAM to do:
 *   - IMUManager::getFactor                 gtsam_fusion/src/gtsam_fusion/IMUManager.cpp:27-74
 *   - PreintegratedCombinedMeasurements     (GTSAM, TangentPreintegration build) [EXTERNAL]
 *   - CombinedImuFactor::evaluateError      constructed at IMUManager.cpp:68-73   [EXTERNAL]
 *   - BetweenFactor<Pose3>::evaluateError   constructed at GraphManager.cpp:86    [EXTERNAL]
 *   - PriorFactor<Pose3/Vector3/ConstantBias> GraphManager.cpp:33-35              [EXTERNAL]
 *   - PreintegrationBase::predict           GraphManager.cpp:153                  [EXTERNAL]
 *   - the solve (GraphManager.cpp:126-129): batch Levenberg-Marquardt on the banded normal
 *     equations -- the north-star algorithm; the reference's live path is iSAM2 (QR).
 * The code deliberately follows GTSAM's function structure (explicit 9x9 Jacobian chains,
 * rotation matrices) so that it is an independent implementation from the HIP kernels,
 * which use closed-form 3x3 block expressions on quaternions.
 *
 * parity unpinned against GTSAM itself for residual/Jacobian values, the preintegrated
 * covariance and the trajectory (GTSAM absent here) -- see header + DESIGN.md.
 */
#include "vf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static double marg_cost(const vfo_marg* m, const double* states, double* grad);
int vfo_marginalize_floor(const vfo_problem* p, int m, double gauge_floor, vfo_marg* out);

/* ------------------------------------------------------------------ small dense helpers */

static void mm(const double* A, const double* B, double* C, int m, int k, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static void mtm(const double* A, const double* B, double* C, int k, int m, int n) {
    /* C(m x n) = A^T B, A is k x m, B is k x n */
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) s += A[l * m + i] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static void mv(const double* A, const double* x, double* y, int m, int n) {
    for (int i = 0; i < m; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += A[i * n + j] * x[j];
        y[i] = s;
    }
}
static void tr3(const double* A, double* At) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) At[j * 3 + i] = A[i * 3 + j];
}
static void skew(const double v[3], double S[9]) {
    S[0] = 0;     S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2];  S[4] = 0;     S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0];  S[8] = 0;
}
static void eye(double* A, int n) {
    memset(A, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) A[i * n + i] = 1.0;
}
static double dot3(const double a[3], const double b[3]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
/* put a 3x3 block into a larger row-major matrix */
static void setblk(double* M, int ld, int r0, int c0, const double* B, int br, int bc, double s) {
    for (int i = 0; i < br; i++)
        for (int j = 0; j < bc; j++) M[(r0 + i) * ld + c0 + j] = s * B[i * bc + j];
}

/* ------------------------------------------------------------------ series coefficients */
/* x = theta^2.  A=sin/th, B=(1-cos)/th^2, C=(th-sin)/th^3, dB=B'(th)/th, dC=C'(th)/th,
 * E = 1/th^2 - (1+cos)/(2 th sin)  (coefficient of W^2 in J_r^{-1}). */
#define SERIES_X 0.25

static void coef_ABC(double x, double* A, double* B, double* C) {
    if (x < SERIES_X) {
        *A = 1.0 + x * (-1.0 / 6 + x * (1.0 / 120 + x * (-1.0 / 5040 + x * (1.0 / 362880 + x * (-1.0 / 39916800 + x * (1.0 / 6227020800.0))))));
        *B = 0.5 + x * (-1.0 / 24 + x * (1.0 / 720 + x * (-1.0 / 40320 + x * (1.0 / 3628800 + x * (-1.0 / 479001600 + x * (1.0 / 87178291200.0))))));
        *C = 1.0 / 6 + x * (-1.0 / 120 + x * (1.0 / 5040 + x * (-1.0 / 362880 + x * (1.0 / 39916800 + x * (-1.0 / 6227020800.0 + x * (1.0 / 1307674368000.0))))));
    } else {
        double th = sqrt(x);
        *A = sin(th) / th;
        *B = (1.0 - cos(th)) / x;
        *C = (1.0 - *A) / x;
    }
}
static void coef_dBdC(double x, double* dB, double* dC) {
    if (x < SERIES_X) {
        *dB = -1.0 / 12 + x * (1.0 / 180 + x * (-1.0 / 6720 + x * (1.0 / 453600 + x * (-1.0 / 47900160 + x * (1.0 / 7264857600.0)))));
        *dC = -1.0 / 60 + x * (1.0 / 1260 + x * (-1.0 / 60480 + x * (1.0 / 4989600 + x * (-1.0 / 622702080 + x * (1.0 / 108972864000.0)))));
    } else {
        double A, B, C;
        coef_ABC(x, &A, &B, &C);
        *dB = (A - 2.0 * B) / x;
        *dC = (B - 3.0 * C) / x;
    }
}
static double coef_E(double x) {
    if (x < SERIES_X) {
        /* sum |B_2n| x^(n-1) / (2n)! */
        return 1.0 / 12 + x * (1.0 / 720 + x * (1.0 / 30240 + x * (1.0 / 1209600 + x * (1.0 / 47900160 + x * (691.0 / 1307674368000.0 + x * (1.0 / 74724249600.0 + x * (3617.0 / 10670622842880000.0)))))));
    }
    double th = sqrt(x);
    return 1.0 / x - (1.0 + cos(th)) / (2.0 * th * sin(th));
}

/* ------------------------------------------------------------------ SO(3) */

void vfo_so3_exp(const double w[3], double R[9]) {
    /* Rot3::Expmap / so3::ExpmapFunctor::expmap: I + A W + B W^2 */
    double x = dot3(w, w), A, B, C, W[9], W2[9];
    coef_ABC(x, &A, &B, &C);
    skew(w, W);
    mm(W, W, W2, 3, 3, 3);
    eye(R, 3);
    for (int i = 0; i < 9; i++) R[i] += A * W[i] + B * W2[i];
}

void vfo_quat_to_rot(const double q[4], double R[9]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double n = w * w + x * x + y * y + z * z;
    double s = 2.0 / n;
    R[0] = 1 - s * (y * y + z * z); R[1] = s * (x * y - w * z);     R[2] = s * (x * z + w * y);
    R[3] = s * (x * y + w * z);     R[4] = 1 - s * (x * x + z * z); R[5] = s * (y * z - w * x);
    R[6] = s * (x * z - w * y);     R[7] = s * (y * z + w * x);     R[8] = 1 - s * (x * x + y * y);
}

void vfo_rot_to_quat(const double R[9], double q[4]) {
    /* Shepperd's method: pick the largest of (trace, R00, R11, R22) */
    double tr = R[0] + R[4] + R[8];
    if (tr > R[0] && tr > R[4] && tr > R[8]) {
        double s = sqrt(1.0 + tr) * 2.0;
        q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2.0;
        q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2.0;
        q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s;
    } else {
        double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2.0;
        q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s;
    }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double sg = q[0] < 0 ? -1.0 / n : 1.0 / n;
    for (int i = 0; i < 4; i++) q[i] *= sg;
}

void vfo_so3_log(const double R[9], double w[3]) {
    /* Rot3::Logmap.  GTSAM's matrix form uses acos(trace); this restatement evaluates the
     * same map through the unit quaternion + atan2, which is well conditioned at small angles. */
    double q[4];
    vfo_rot_to_quat(R, q);
    double n = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double f;
    if (n < 1e-7) {
        double r = n / q[0];
        f = 2.0 / q[0] * (1.0 - r * r / 3.0);
    } else {
        f = 2.0 * atan2(n, q[0]) / n;
    }
    w[0] = f * q[1]; w[1] = f * q[2]; w[2] = f * q[3];
}

void vfo_so3_jr(const double w[3], double J[9]) {
    /* Rot3::ExpmapDerivative = so3::DexpFunctor::dexp = I - B W + C W^2 */
    double x = dot3(w, w), A, B, C, W[9], W2[9];
    coef_ABC(x, &A, &B, &C);
    skew(w, W);
    mm(W, W, W2, 3, 3, 3);
    eye(J, 3);
    for (int i = 0; i < 9; i++) J[i] += -B * W[i] + C * W2[i];
}

void vfo_so3_jr_inv(const double w[3], double J[9]) {
    /* Rot3::LogmapDerivative = I + W/2 + E W^2 */
    double x = dot3(w, w), E = coef_E(x), W[9], W2[9];
    skew(w, W);
    mm(W, W, W2, 3, 3, 3);
    eye(J, 3);
    for (int i = 0; i < 9; i++) J[i] += 0.5 * W[i] + E * W2[i];
}

/* d/dtheta [ J_r(theta) c ] for fixed c   (so3::DexpFunctor::applyDexp H1) */
static void so3_jr_apply_dtheta(const double th[3], const double c[3], double D[9]) {
    double x = dot3(th, th), A, B, C, dB, dC;
    coef_ABC(x, &A, &B, &C);
    coef_dBdC(x, &dB, &dC);
    double txc[3], ttxc[3], Sc[9];
    cross3(th, c, txc);
    cross3(th, txc, ttxc);
    skew(c, Sc);
    double tc = dot3(th, c);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double v = B * Sc[i * 3 + j] - txc[i] * dB * th[j] + ttxc[i] * dC * th[j] +
                       C * (th[i] * c[j] - 2.0 * c[i] * th[j]);
            if (i == j) v += C * tc;
            D[i * 3 + j] = v;
        }
}

/* ------------------------------------------------------------------ SE(3) */

void vfo_se3_exp(const double xi[6], double R[9], double t[3]) {
    /* Pose3::Expmap: (Exp(w), J_l(w) v) */
    double x = dot3(xi, xi), A, B, C, W[9], W2[9], V[9];
    coef_ABC(x, &A, &B, &C);
    vfo_so3_exp(xi, R);
    skew(xi, W);
    mm(W, W, W2, 3, 3, 3);
    eye(V, 3);
    for (int i = 0; i < 9; i++) V[i] += B * W[i] + C * W2[i];
    mv(V, xi + 3, t, 3, 3);
}

void vfo_se3_log(const double R[9], const double t[3], double xi[6]) {
    /* Pose3::Logmap: w = Log(R), u = J_l(w)^{-1} t = (I - W/2 + E W^2) t */
    vfo_so3_log(R, xi);
    double x = dot3(xi, xi), E = coef_E(x), W[9], W2[9], Vi[9];
    skew(xi, W);
    mm(W, W, W2, 3, 3, 3);
    eye(Vi, 3);
    for (int i = 0; i < 9; i++) Vi[i] += -0.5 * W[i] + E * W2[i];
    mv(Vi, t, xi + 3, 3, 3);
}

/* Barfoot's Q(xi) for the LEFT Jacobian, [w,v] ordering: J_l = [[Jl(w),0],[Q,Jl(w)]] */
static void se3_Q_left(const double w[3], const double v[3], double Q[9]) {
    double x = dot3(w, w), A, B, C, c1, c2, c3;
    coef_ABC(x, &A, &B, &C);
    c1 = C;
    if (x < SERIES_X) {
        c2 = 1.0 / 24 + x * (-1.0 / 720 + x * (1.0 / 40320 + x * (-1.0 / 3628800 + x * (1.0 / 479001600 + x * (-1.0 / 87178291200.0)))));
        c3 = 1.0 / 120 + x * (-1.0 / 2520 + x * (1.0 / 120960 + x * (-1.0 / 9979200 + x * (1.0 / 1245404160.0 + x * (-1.0 / 217945728000.0)))));
    } else {
        double th = sqrt(x);
        c2 = (x + 2.0 * cos(th) - 2.0) / (2.0 * x * x);
        c3 = (2.0 * th - 3.0 * sin(th) + th * cos(th)) / (2.0 * x * x * th);
    }
    double W[9], V[9], WV[9], VW[9], WVW[9], WWV[9], VWW[9], WVWW[9], WWVW[9], WW[9];
    skew(w, W);
    skew(v, V);
    mm(W, V, WV, 3, 3, 3);
    mm(V, W, VW, 3, 3, 3);
    mm(WV, W, WVW, 3, 3, 3);
    mm(W, W, WW, 3, 3, 3);
    mm(WW, V, WWV, 3, 3, 3);
    mm(V, WW, VWW, 3, 3, 3);
    mm(WVW, W, WVWW, 3, 3, 3);
    mm(W, WVW, WWVW, 3, 3, 3);
    for (int i = 0; i < 9; i++)
        Q[i] = 0.5 * V[i] + c1 * (WV[i] + VW[i] + WVW[i]) + c2 * (WWV[i] + VWW[i] - 3.0 * WVW[i]) +
               c3 * (WVWW[i] + WWVW[i]);
}

void vfo_se3_jr_inv(const double xi[6], double J[36]) {
    /* Pose3::LogmapDerivative: J_r^{-1}(xi) = J_l^{-1}(-xi) = [[Jw,0],[-Jw Q Jw, Jw]],
     * Jw = J_r^{-1}(w), Q = Q_left(-xi). */
    double Jw[9], Q[9], nw[3] = {-xi[0], -xi[1], -xi[2]}, nv[3] = {-xi[3], -xi[4], -xi[5]};
    double T1[9], T2[9];
    vfo_so3_jr_inv(xi, Jw);
    se3_Q_left(nw, nv, Q);
    mm(Jw, Q, T1, 3, 3, 3);
    mm(T1, Jw, T2, 3, 3, 3);
    memset(J, 0, sizeof(double) * 36);
    setblk(J, 6, 0, 0, Jw, 3, 3, 1.0);
    setblk(J, 6, 3, 3, Jw, 3, 3, 1.0);
    setblk(J, 6, 3, 0, T2, 3, 3, -1.0);
}

/* ------------------------------------------------------------------ preintegration */

void vfo_pim_reset(vfo_pim* p, const double bhat[6]) {
    /* PreintegrationBase::resetIntegrationAndSetBias (IMUManager.cpp:42) */
    memset(p, 0, sizeof(*p));
    memcpy(p->bhat, bhat, sizeof(double) * 6);
}

void vfo_pim_integrate(vfo_pim* p, const vfo_imu_params* prm, const double macc[3],
                       const double mgyro[3], double dt) {
    /* PreintegratedCombinedMeasurements::integrateMeasurement (IMUManager.cpp:50,64):
     * TangentPreintegration::update + 15x15 covariance propagation (GTSAM 4.0.x form). */
    double acc[3], om[3];
    for (int i = 0; i < 3; i++) {
        acc[i] = macc[i] - p->bhat[i];
        om[i] = mgyro[i] - p->bhat[3 + i];
    }
    const double* th = p->d;
    double Jr[9], invD[9], R[9], wt[3], anav[3];
    vfo_so3_jr(th, Jr);
    vfo_so3_jr_inv(th, invD);
    mv(invD, om, wt, 3, 3);
    vfo_so3_exp(th, R);
    mv(R, acc, anav, 3, 3);
    const double dt22 = 0.5 * dt * dt;

    /* A (9x9), B (9x3), C (9x3) of UpdatePreintegrated */
    double Am[81], Bm[27], Cm[27];
    double Ddexp[9], wH[9], T[9], Sa[9], aH[9], nacc[3] = {-acc[0], -acc[1], -acc[2]};
    so3_jr_apply_dtheta(th, wt, Ddexp);
    mm(invD, Ddexp, wH, 3, 3, 3); /* w_tangent_H_theta = -invDexp * D_dexpv_omega */
    skew(nacc, Sa);
    mm(R, Sa, T, 3, 3, 3);
    mm(T, Jr, aH, 3, 3, 3); /* a_nav_H_theta */
    eye(Am, 9);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            Am[i * 9 + j] += -wH[i * 3 + j] * dt;
            Am[(3 + i) * 9 + j] = aH[i * 3 + j] * dt22;
            Am[(6 + i) * 9 + j] = aH[i * 3 + j] * dt;
        }
    for (int i = 0; i < 3; i++) Am[(3 + i) * 9 + 6 + i] = dt;
    memset(Bm, 0, sizeof(Bm));
    memset(Cm, 0, sizeof(Cm));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            Bm[(3 + i) * 3 + j] = R[i * 3 + j] * dt22;
            Bm[(6 + i) * 3 + j] = R[i * 3 + j] * dt;
            Cm[i * 3 + j] = invD[i * 3 + j] * dt;
        }

    /* mean */
    double nd[9];
    for (int i = 0; i < 3; i++) {
        nd[i] = p->d[i] + wt[i] * dt;
        nd[3 + i] = p->d[3 + i] + p->d[6 + i] * dt + anav[i] * dt22;
        nd[6 + i] = p->d[6 + i] + anav[i] * dt;
    }
    memcpy(p->d, nd, sizeof(nd));
    p->dt += dt;

    /* bias Jacobians: H_acc = A H_acc - B ; H_omega = A H_omega - C */
    double Hn[54];
    mm(Am, p->H, Hn, 9, 9, 6);
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 3; j++) {
            Hn[i * 6 + j] -= Bm[i * 3 + j];
            Hn[i * 6 + 3 + j] -= Cm[i * 3 + j];
        }
    memcpy(p->H, Hn, sizeof(Hn));

    /* covariance: F P F^T + G Q G^T (block form of the 4.0.x source) */
    double F[225];
    memset(F, 0, sizeof(F));
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) F[i * 15 + j] = Am[i * 9 + j];
    double tHb[9], vHb[9]; /* theta_H_biasOmega = -C.top, vel_H_biasAcc = -B.bottom */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            tHb[i * 3 + j] = -Cm[i * 3 + j];
            vHb[i * 3 + j] = -Bm[(6 + i) * 3 + j];
            F[i * 15 + 12 + j] = tHb[i * 3 + j];
            F[(6 + i) * 15 + 9 + j] = vHb[i * 3 + j];
        }
    for (int i = 9; i < 15; i++) F[i * 15 + i] = 1.0;
    double FP[225], FPFt[225];
    mm(F, p->cov, FP, 15, 15, 15);
    for (int i = 0; i < 15; i++)
        for (int j = 0; j < 15; j++) {
            double s = 0.0;
            for (int l = 0; l < 15; l++) s += FP[i * 15 + l] * F[j * 15 + l];
            FPFt[i * 15 + j] = s;
        }
    double G[225];
    memset(G, 0, sizeof(G));
    double vv[9], rr[9], tmp[9], vHbT[9], tHbT[9];
    tr3(vHb, vHbT);
    tr3(tHb, tHbT);
    mm(vHb, vHbT, vv, 3, 3, 3);
    mm(tHb, tHbT, rr, 3, 3, 3);
    const double sv = (prm->acc_cov + prm->bias_acc_omega_int) / dt;
    const double sr = (prm->gyro_cov + prm->bias_acc_omega_int) / dt;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            G[(6 + i) * 15 + 6 + j] = sv * vv[i * 3 + j]; /* D_v_v */
            G[i * 15 + j] = sr * rr[i * 3 + j];           /* D_R_R */
        }
        G[(3 + i) * 15 + 3 + i] = dt * prm->int_cov;          /* D_t_t */
        G[(9 + i) * 15 + 9 + i] = dt * prm->bias_acc_cov;     /* D_a_a */
        G[(12 + i) * 15 + 12 + i] = dt * prm->bias_omega_cov; /* D_g_g */
    }
    /* off-diagonal D_v_R uses biasAccOmegaInt.block<3,3>(3,0), which is zero for
     * Matrix6::Identity()*c (ImuManagerRos.cpp:33) */
    (void)tmp;
    for (int i = 0; i < 225; i++) p->cov[i] = FPFt[i] + G[i];
}

int vfo_imu_get_factor(const double* t, const double* acc, const double* gyro, int n, int* head,
                       double start, double end, const double bias[6], const vfo_imu_params* prm,
                       vfo_pim* out) {
    /* IMUManager::getFactor, IMUManager.cpp:27-74.  Returns #integrations performed. */
    int h = *head, count = 0;
    double pt = 0, pa[3] = {0, 0, 0}, pg[3] = {0, 0, 0};
    while (h < n && t[h] <= start) { /* :35-40 drop old samples */
        pt = t[h];
        memcpy(pa, acc + 3 * h, sizeof(pa));
        memcpy(pg, gyro + 3 * h, sizeof(pg));
        h++;
    }
    vfo_pim_reset(out, bias); /* :42 */
    pt = start;               /* :44 */
    while (h < n && t[h] < end) { /* :46-54 */
        vfo_pim_integrate(out, prm, acc + 3 * h, gyro + 3 * h, t[h] - pt);
        pt = t[h];
        memcpy(pa, acc + 3 * h, sizeof(pa));
        memcpy(pg, gyro + 3 * h, sizeof(pg));
        h++;
        count++;
    }
    if (h < n) { /* :57-66 interpolate the final sample; it stays in the buffer */
        double f = (end - pt) / (t[h] - pt), ia[3], ig[3];
        for (int i = 0; i < 3; i++) {
            ia[i] = f * acc[3 * h + i] + (1.0 - f) * pa[i];
            ig[i] = f * gyro[3 * h + i] + (1.0 - f) * pg[i];
        }
        vfo_pim_integrate(out, prm, ia, ig, end - pt);
        count++;
    }
    *head = h;
    return count;
}

/* Gauss-Jordan inverse with partial pivoting (Eigen's covariance.inverse() is LU based). */
static int inv_gj(const double* A, int n, double* Ai) {
    double* M = (double*)malloc(sizeof(double) * n * 2 * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            M[i * 2 * n + j] = A[i * n + j];
            M[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; c++) {
        int pr = c;
        for (int r = c + 1; r < n; r++)
            if (fabs(M[r * 2 * n + c]) > fabs(M[pr * 2 * n + c])) pr = r;
        if (M[pr * 2 * n + c] == 0.0) { free(M); return -1; }
        if (pr != c)
            for (int j = 0; j < 2 * n; j++) {
                double t = M[c * 2 * n + j]; M[c * 2 * n + j] = M[pr * 2 * n + j]; M[pr * 2 * n + j] = t;
            }
        double ip = 1.0 / M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; j++) M[c * 2 * n + j] *= ip;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            double f = M[r * 2 * n + c];
            if (f != 0.0)
                for (int j = 0; j < 2 * n; j++) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ai[i * n + j] = M[i * 2 * n + n + j];
    free(M);
    return 0;
}

int vfo_sqrt_info_upper(const double* cov, int n, double* Rp) {
    /* noiseModel::Gaussian::Covariance(cov) -> Information(cov.inverse()) -> LLT.matrixU()
     * (used by CombinedImuFactor's ctor and SensorManagerRos.cpp:99) */
    double* I = (double*)malloc(sizeof(double) * n * n);
    double* L = (double*)calloc((size_t)n * n, sizeof(double));
    if (inv_gj(cov, n, I) != 0) { free(I); free(L); return -1; }
    for (int i = 0; i < n; i++) /* symmetrise */
        for (int j = 0; j < i; j++) {
            double s = 0.5 * (I[i * n + j] + I[j * n + i]);
            I[i * n + j] = I[j * n + i] = s;
        }
    for (int j = 0; j < n; j++) { /* lower Cholesky I = L L^T, R = L^T */
        double s = I[j * n + j];
        for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
        if (!(s > 0.0)) { free(I); free(L); return -1; }
        L[j * n + j] = sqrt(s);
        for (int i = j + 1; i < n; i++) {
            double v = I[i * n + j];
            for (int k = 0; k < j; k++) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = v / L[j * n + j];
        }
    }
    int o = 0;
    for (int r = 0; r < n; r++)
        for (int c = r; c < n; c++) Rp[o++] = L[c * n + r];
    free(I);
    free(L);
    return 0;
}

int vfo_pim_to_record(const vfo_pim* p, double rec[VFO_IMU_DATA]) {
    rec[0] = p->dt;
    memcpy(rec + 1, p->d, sizeof(double) * 9);
    memcpy(rec + 10, p->bhat, sizeof(double) * 6);
    memcpy(rec + 16, p->H, sizeof(double) * 54);
    return vfo_sqrt_info_upper(p->cov, 15, rec + 70);
}

/* ------------------------------------------------------------------ NavState algebra */

typedef struct { double R[9], t[3], v[3]; } navstate;

static void nav_from_state(const double x[16], navstate* s) {
    vfo_quat_to_rot(x, s->R);
    memcpy(s->t, x + 4, sizeof(double) * 3);
    memcpy(s->v, x + 7, sizeof(double) * 3);
}

/* TangentPreintegration::biasCorrectedDelta */
static void bias_corrected_delta(const double rec[VFO_IMU_DATA], const double bias[6], double bc[9]) {
    const double* d = rec + 1;
    const double* bhat = rec + 10;
    const double* H = rec + 16;
    double inc[6];
    for (int i = 0; i < 6; i++) inc[i] = bias[i] - bhat[i];
    for (int i = 0; i < 9; i++) {
        double s = d[i];
        for (int j = 0; j < 6; j++) s += H[i * 6 + j] * inc[j];
        bc[i] = s;
    }
}

/* NavState::correctPIM (no Coriolis). H1 (9x9) wrt state tangent; H2 = I. */
static void nav_correct_pim(const navstate* s, const double pim[9], double dt, const double g[3],
                            double xi[9], double* H1) {
    double Rt[9], rv[3], rg[3];
    tr3(s->R, Rt);
    mv(Rt, s->v, rv, 3, 3);
    mv(Rt, g, rg, 3, 3);
    const double dt22 = 0.5 * dt * dt;
    for (int i = 0; i < 3; i++) {
        xi[i] = pim[i];
        xi[3 + i] = pim[3 + i] + dt * rv[i] + dt22 * rg[i];
        xi[6 + i] = pim[6 + i] + dt * rg[i];
    }
    if (H1) {
        double Sv[9], Sg[9];
        skew(rv, Sv); /* Rot3::unrotate H1 = skew(R^T p) */
        skew(rg, Sg);
        memset(H1, 0, sizeof(double) * 81);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                H1[(3 + i) * 9 + j] = dt * Sv[i * 3 + j] + dt22 * Sg[i * 3 + j];
                H1[(6 + i) * 9 + j] = dt * Sg[i * 3 + j];
            }
        for (int i = 0; i < 3; i++) H1[(3 + i) * 9 + 6 + i] = dt; /* dt * R^T * R */
    }
}

/* NavState::retract. H1 wrt this, H2 wrt xi. */
static void nav_retract(const navstate* s, const double xi[9], navstate* out, double* H1, double* H2) {
    double bRc[9], bRcT[9], rp[3], rv[3];
    vfo_so3_exp(xi, bRc);
    mm(s->R, bRc, out->R, 3, 3, 3);
    mv(s->R, xi + 3, rp, 3, 3);
    mv(s->R, xi + 6, rv, 3, 3);
    for (int i = 0; i < 3; i++) {
        out->t[i] = s->t[i] + rp[i];
        out->v[i] = s->v[i] + rv[i];
    }
    tr3(bRc, bRcT);
    if (H1) {
        double Sp[9], Sv[9], T1[9], T2[9];
        skew(xi + 3, Sp);
        skew(xi + 6, Sv);
        mm(bRcT, Sp, T1, 3, 3, 3); /* nRc^T * (nRb * skew(-p)) = -bRc^T skew(p) */
        mm(bRcT, Sv, T2, 3, 3, 3);
        memset(H1, 0, sizeof(double) * 81);
        setblk(H1, 9, 0, 0, bRcT, 3, 3, 1.0);
        setblk(H1, 9, 3, 0, T1, 3, 3, -1.0);
        setblk(H1, 9, 3, 3, bRcT, 3, 3, 1.0);
        setblk(H1, 9, 6, 0, T2, 3, 3, -1.0);
        setblk(H1, 9, 6, 6, bRcT, 3, 3, 1.0);
    }
    if (H2) {
        double Jr[9];
        vfo_so3_jr(xi, Jr);
        memset(H2, 0, sizeof(double) * 81);
        setblk(H2, 9, 0, 0, Jr, 3, 3, 1.0);
        setblk(H2, 9, 3, 3, bRcT, 3, 3, 1.0);
        setblk(H2, 9, 6, 6, bRcT, 3, 3, 1.0);
    }
}

/* NavState::localCoordinates(g). H1 wrt this, H2 wrt g. */
static void nav_local(const navstate* s, const navstate* g, double xi[9], double* H1, double* H2) {
    double Rt[9], dR[9], dtv[3], dvv[3], d1[3], d2[3];
    tr3(s->R, Rt);
    mm(Rt, g->R, dR, 3, 3, 3);
    for (int i = 0; i < 3; i++) {
        d1[i] = g->t[i] - s->t[i];
        d2[i] = g->v[i] - s->v[i];
    }
    mv(Rt, d1, dtv, 3, 3);
    mv(Rt, d2, dvv, 3, 3);
    vfo_so3_log(dR, xi);
    memcpy(xi + 3, dtv, sizeof(dtv));
    memcpy(xi + 6, dvv, sizeof(dvv));
    if (H1 || H2) {
        double L[9];
        vfo_so3_jr_inv(xi, L); /* D_xi_R = Rot3::LogmapDerivative */
        if (H1) {
            double dRt[9], T[9], St[9], Sv[9];
            tr3(dR, dRt);
            mm(L, dRt, T, 3, 3, 3); /* D_xi_R * D_dR_R, D_dR_R = -dR^T */
            skew(dtv, St);
            skew(dvv, Sv);
            memset(H1, 0, sizeof(double) * 81);
            setblk(H1, 9, 0, 0, T, 3, 3, -1.0);
            setblk(H1, 9, 3, 0, St, 3, 3, 1.0);
            setblk(H1, 9, 6, 0, Sv, 3, 3, 1.0);
            for (int i = 3; i < 9; i++) H1[i * 9 + i] = -1.0;
        }
        if (H2) {
            memset(H2, 0, sizeof(double) * 81);
            setblk(H2, 9, 0, 0, L, 3, 3, 1.0);
            setblk(H2, 9, 3, 3, dR, 3, 3, 1.0);
            setblk(H2, 9, 6, 6, dR, 3, 3, 1.0);
        }
    }
}

/* PreintegrationBase::predict with Jacobians wrt state_i (9x9) and bias (9x6) */
static void pim_predict(const double rec[VFO_IMU_DATA], const double g[3], const navstate* si,
                        const double bias[6], navstate* sj, double* H1, double* H2) {
    double bc[9], xi[9], Dds[81], Dps[81], Dpd[81];
    bias_corrected_delta(rec, bias, bc);
    nav_correct_pim(si, bc, rec[0], g, xi, (H1) ? Dds : NULL);
    nav_retract(si, xi, sj, (H1) ? Dps : NULL, (H1 || H2) ? Dpd : NULL);
    if (H1) {
        double T[81];
        mm(Dpd, Dds, T, 9, 9, 9);
        for (int i = 0; i < 81; i++) H1[i] = Dps[i] + T[i];
    }
    if (H2) mm(Dpd, rec + 16, H2, 9, 9, 6); /* D_predict_delta * I * [H_acc H_omega] */
}

void vfo_predict(const double rec[VFO_IMU_DATA], const double gravity[3], const double xi[16],
                 double xj[16]) {
    /* GraphManager::emptyImuQueue (GraphManager.cpp:152-160): pose/vel from predict, bias copied */
    navstate si, sj;
    nav_from_state(xi, &si);
    pim_predict(rec, gravity, &si, xi + 10, &sj, NULL, NULL);
    vfo_rot_to_quat(sj.R, xj);
    memcpy(xj + 4, sj.t, sizeof(double) * 3);
    memcpy(xj + 7, sj.v, sizeof(double) * 3);
    memcpy(xj + 10, xi + 10, sizeof(double) * 6);
}

/* unpack packed upper-triangular (row-major) into dense n x n */
static void unpack_upper(const double* Rp, int n, double* R) {
    memset(R, 0, sizeof(double) * n * n);
    int o = 0;
    for (int r = 0; r < n; r++)
        for (int c = r; c < n; c++) R[r * n + c] = Rp[o++];
}

void vfo_imu_factor(const double rec[VFO_IMU_DATA], const double gravity[3], const double xi[16],
                    const double xj[16], int whiten, double r[15], double J[450]) {
    /* CombinedImuFactor::evaluateError -> PreintegrationBase::computeErrorAndJacobians
     * -> computeError -> predict + localCoordinates. */
    navstate si, sj, pj;
    nav_from_state(xi, &si);
    nav_from_state(xj, &sj);
    double Dpi[81], Dpb[54], Dej[81], Dep[81];
    pim_predict(rec, gravity, &si, xi + 10, &pj, Dpi, Dpb);
    double e9[9];
    nav_local(&sj, &pj, e9, Dej, Dep);
    double Dei[81], Deb[54];
    mm(Dep, Dpi, Dei, 9, 9, 9); /* D_error_state_i */
    mm(Dep, Dpb, Deb, 9, 9, 6); /* D_error_bias_i  */

    double Ju[450];
    memset(Ju, 0, sizeof(Ju));
    double RiT[9], RjT[9];
    tr3(si.R, RiT);
    tr3(sj.R, RjT);
    for (int i = 0; i < 9; i++) {
        for (int j = 0; j < 6; j++) {
            Ju[i * 30 + j] = Dei[i * 9 + j];      /* H1 pose_i */
            Ju[i * 30 + 9 + j] = Dej[i * 9 + j];  /* H3 pose_j */
            Ju[i * 30 + 18 + j] = Deb[i * 6 + j]; /* H5 bias_i (top 9 rows) */
        }
        for (int j = 0; j < 3; j++) {
            double s1 = 0, s2 = 0;
            for (int l = 0; l < 3; l++) {
                s1 += Dei[i * 9 + 6 + l] * RiT[l * 3 + j]; /* H2 = rightCols<3> * R_i^T */
                s2 += Dej[i * 9 + 6 + l] * RjT[l * 3 + j]; /* H4 */
            }
            Ju[i * 30 + 6 + j] = s1;
            Ju[i * 30 + 15 + j] = s2;
        }
    }
    /* bias random walk: fbias = Between(bias_j, bias_i) = bias_i - bias_j */
    double ru[15];
    memcpy(ru, e9, sizeof(e9));
    for (int i = 0; i < 6; i++) {
        ru[9 + i] = xi[10 + i] - xj[10 + i];
        Ju[(9 + i) * 30 + 18 + i] = 1.0;
        Ju[(9 + i) * 30 + 24 + i] = -1.0;
    }
    if (!whiten) {
        memcpy(r, ru, sizeof(ru));
        memcpy(J, Ju, sizeof(Ju));
        return;
    }
    double R[225];
    unpack_upper(rec + 70, 15, R);
    mv(R, ru, r, 15, 15);
    mm(R, Ju, J, 15, 15, 30);
}

void vfo_between_factor(const double rec[VFO_BTW_DATA], const double xa[16], const double xb[16],
                        int whiten, double r[6], double Ja[36], double Jb[36]) {
    /* BetweenFactor<Pose3>::evaluateError: hx = between(p1,p2) (H1=-Ad(hx^-1), H2=I);
     * rval = Local(measured, hx) = Logmap(measured^-1 hx), Hlocal = LogmapDerivative. */
    double Ra[9], Rb[9], Rm[9], RaT[9], RmT[9], Rh[9], th[3], d[3];
    vfo_quat_to_rot(xa, Ra);
    vfo_quat_to_rot(xb, Rb);
    vfo_quat_to_rot(rec, Rm);
    tr3(Ra, RaT);
    tr3(Rm, RmT);
    mm(RaT, Rb, Rh, 3, 3, 3);
    for (int i = 0; i < 3; i++) d[i] = xb[4 + i] - xa[4 + i];
    mv(RaT, d, th, 3, 3); /* hx = (Rh, th) */
    double Re[9], te[3], d2[3];
    mm(RmT, Rh, Re, 3, 3, 3);
    for (int i = 0; i < 3; i++) d2[i] = th[i] - rec[4 + i];
    mv(RmT, d2, te, 3, 3); /* measured^-1 * hx */
    double ru[6], Hl[36];
    vfo_se3_log(Re, te, ru);
    vfo_se3_jr_inv(ru, Hl);
    /* Ad(hx^-1): hx^-1 = (Rh^T, -Rh^T th); Ad(T) = [[R,0],[skew(t) R, R]] */
    double RhT[9], ti[3], St[9], StR[9], Ad[36];
    tr3(Rh, RhT);
    mv(RhT, th, ti, 3, 3);
    for (int i = 0; i < 3; i++) ti[i] = -ti[i];
    skew(ti, St);
    mm(St, RhT, StR, 3, 3, 3);
    memset(Ad, 0, sizeof(Ad));
    setblk(Ad, 6, 0, 0, RhT, 3, 3, 1.0);
    setblk(Ad, 6, 3, 3, RhT, 3, 3, 1.0);
    setblk(Ad, 6, 3, 0, StR, 3, 3, 1.0);
    double Jau[36], Jbu[36];
    mm(Hl, Ad, Jau, 6, 6, 6);
    for (int i = 0; i < 36; i++) {
        Jau[i] = -Jau[i];
        Jbu[i] = Hl[i];
    }
    if (!whiten) {
        memcpy(r, ru, sizeof(ru));
        memcpy(Ja, Jau, sizeof(Jau));
        memcpy(Jb, Jbu, sizeof(Jbu));
        return;
    }
    double R[36];
    unpack_upper(rec + 7, 6, R);
    mv(R, ru, r, 6, 6);
    mm(R, Jau, Ja, 6, 6, 6);
    mm(R, Jbu, Jb, 6, 6, 6);
}

void vfo_prior_factor(const double rec[VFO_PRIOR_DATA], const double x[16], double r[15],
                      double J[225]) {
    /* PriorFactor<Pose3>: Local(prior, x) = Logmap(prior^-1 x), H = LogmapDerivative;
     * PriorFactor<Vector3>, PriorFactor<ConstantBias>: x - prior, H = I.  Diagonal sigmas
     * (GraphManager.cpp:27-35). */
    double Rp[9], Rx[9], RpT[9], Re[9], d[3], te[3], xi[6], Hl[36];
    vfo_quat_to_rot(rec, Rp);
    vfo_quat_to_rot(x, Rx);
    tr3(Rp, RpT);
    mm(RpT, Rx, Re, 3, 3, 3);
    for (int i = 0; i < 3; i++) d[i] = x[4 + i] - rec[4 + i];
    mv(RpT, d, te, 3, 3);
    vfo_se3_log(Re, te, xi);
    vfo_se3_jr_inv(xi, Hl);
    const double* sig = rec + 16;
    memset(J, 0, sizeof(double) * 225);
    for (int i = 0; i < 6; i++) {
        r[i] = xi[i] / sig[i];
        for (int j = 0; j < 6; j++) J[i * 15 + j] = Hl[i * 6 + j] / sig[i];
    }
    for (int i = 6; i < 15; i++) {
        r[i] = (x[1 + i] - rec[1 + i]) / sig[i]; /* state index = tangent index + 1 for v, b */
        J[i * 15 + i] = 1.0 / sig[i];
    }
}

void vfo_retract(const double x[16], const double delta[15], double out[16]) {
    /* Values::retract: Pose3 (full Expmap chart) x * Exp(d); Vector3 / ConstantBias add. */
    double R[9], dR[9], dtv[3], Rn[9], rt[3];
    vfo_quat_to_rot(x, R);
    vfo_se3_exp(delta, dR, dtv);
    mm(R, dR, Rn, 3, 3, 3);
    mv(R, dtv, rt, 3, 3);
    vfo_rot_to_quat(Rn, out);
    for (int i = 0; i < 3; i++) out[4 + i] = x[4 + i] + rt[i];
    for (int i = 0; i < 9; i++) out[7 + i] = x[7 + i] + delta[6 + i];
}

/* ------------------------------------------------------------------ marginal prior */

/* tangent index map of the 27-vector: keyframe offset and dof */
static void marg_index(int i, int* kf, int* dof) {
    if (i < 15) { *kf = 0; *dof = i; }
    else if (i < 21) { *kf = 1; *dof = i - 15; }
    else { *kf = 2; *dof = i - 21; }
}

void vfo_marg_delta(const vfo_marg* m, const double* states, double d[27]) {
    for (int j = 0; j < 3; j++) {
        const double* xb = m->xbar + 16 * j;
        const double* x = states + 16 * (m->k0 + j);
        double Rb[9], Rx[9], RbT[9], Re[9], dt[3], te[3], xi[6];
        vfo_quat_to_rot(xb, Rb);
        vfo_quat_to_rot(x, Rx);
        tr3(Rb, RbT);
        mm(RbT, Rx, Re, 3, 3, 3);
        for (int i = 0; i < 3; i++) dt[i] = x[4 + i] - xb[4 + i];
        mv(RbT, dt, te, 3, 3);
        vfo_se3_log(Re, te, xi);   /* Local(xbar, x) for the Pose3 full-Expmap chart */
        if (j == 0) {
            memcpy(d, xi, sizeof(double) * 6);
            for (int i = 0; i < 9; i++) d[6 + i] = x[7 + i] - xb[7 + i];
        } else {
            memcpy(d + 15 + 6 * (j - 1), xi, sizeof(double) * 6);
        }
    }
}

/* cost 0.5 d^T L d + eta^T d ; optionally the gradient L d + eta */
static double marg_cost(const vfo_marg* m, const double* states, double* grad) {
    double d[27], g[27];
    vfo_marg_delta(m, states, d);
    mv(m->L, d, g, 27, 27);
    double c = 0.0;
    for (int i = 0; i < 27; i++) {
        c += 0.5 * d[i] * g[i] + m->eta[i] * d[i];
        g[i] += m->eta[i];
    }
    if (grad) memcpy(grad, g, sizeof(g));
    return c;
}

/* ------------------------------------------------------------------ window problem */

int vfo_bandwidth(const vfo_problem* p) {
    int w = 0;
    for (int f = 0; f < p->n_imu; f++) {
        int d = abs(p->imu_j[f] - p->imu_i[f]);
        if (d > w) w = d;
    }
    for (int f = 0; f < p->n_btw; f++) {
        int d = abs(p->btw_b[f] - p->btw_a[f]);
        if (d > w) w = d;
    }
    if (p->marg && p->marg->on && w < 2) w = 2;   /* the marginal prior couples k0 and k0+2 */
    return w;
}

double vfo_cost(const vfo_problem* p) {
    double c = 0.0;
    for (int f = 0; f < p->n_imu; f++) {
        double r[15], J[450];
        vfo_imu_factor(p->imu_data + (size_t)f * VFO_IMU_DATA, p->gravity,
                       p->states + 16 * p->imu_i[f], p->states + 16 * p->imu_j[f], 1, r, J);
        for (int i = 0; i < 15; i++) c += 0.5 * r[i] * r[i];
    }
    for (int f = 0; f < p->n_btw; f++) {
        double r[6], Ja[36], Jb[36];
        vfo_between_factor(p->btw_data + (size_t)f * VFO_BTW_DATA, p->states + 16 * p->btw_a[f],
                           p->states + 16 * p->btw_b[f], 1, r, Ja, Jb);
        for (int i = 0; i < 6; i++) c += 0.5 * r[i] * r[i];
    }
    for (int f = 0; f < p->n_prior; f++) {
        double r[15], J[225];
        vfo_prior_factor(p->prior_data + (size_t)f * VFO_PRIOR_DATA, p->states + 16 * p->prior_k[f], r, J);
        for (int i = 0; i < 15; i++) c += 0.5 * r[i] * r[i];
    }
    if (p->marg && p->marg->on) c += marg_cost(p->marg, p->states, NULL);
    return c;
}

/* column map: GTSAM key order of the 15x30 IMU Jacobian -> per-keyframe tangent order */
static const int IMU_COL_I[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 18, 19, 20, 21, 22, 23};
static const int IMU_COL_J[15] = {9, 10, 11, 12, 13, 14, 15, 16, 17, 24, 25, 26, 27, 28, 29};

#define HB(k, d) (Hband + ((size_t)(k) * (w + 1) + (d)) * 225)

double vfo_assemble(const vfo_problem* p, int w, double* Hband, double* g, int n_threads) {
    const int n = p->n_kf;
    memset(Hband, 0, sizeof(double) * (size_t)n * (w + 1) * 225);
    memset(g, 0, sizeof(double) * (size_t)n * 15);
    double cost = 0.0;
    /* linearise IMU factors (optionally in parallel), then scatter sequentially */
    double* rJ = (double*)malloc(sizeof(double) * (size_t)(p->n_imu > 0 ? p->n_imu : 1) * 465);
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static)
#endif
    for (int f = 0; f < p->n_imu; f++) {
        vfo_imu_factor(p->imu_data + (size_t)f * VFO_IMU_DATA, p->gravity,
                       p->states + 16 * p->imu_i[f], p->states + 16 * p->imu_j[f], 1,
                       rJ + (size_t)f * 465, rJ + (size_t)f * 465 + 15);
    }
    (void)n_threads;
    for (int f = 0; f < p->n_imu; f++) {
        const double* r = rJ + (size_t)f * 465;
        const double* J = r + 15;
        int i = p->imu_i[f], j = p->imu_j[f];
        double Ji[225], Jj[225], T[225];
        for (int a = 0; a < 15; a++)
            for (int b = 0; b < 15; b++) {
                Ji[a * 15 + b] = J[a * 30 + IMU_COL_I[b]];
                Jj[a * 15 + b] = J[a * 30 + IMU_COL_J[b]];
            }
        for (int a = 0; a < 15; a++) cost += 0.5 * r[a] * r[a];
        mtm(Ji, Ji, T, 15, 15, 15);
        for (int a = 0; a < 225; a++) HB(i, 0)[a] += T[a];
        mtm(Jj, Jj, T, 15, 15, 15);
        for (int a = 0; a < 225; a++) HB(j, 0)[a] += T[a];
        if (j > i) {
            mtm(Jj, Ji, T, 15, 15, 15);
            for (int a = 0; a < 225; a++) HB(j, j - i)[a] += T[a];
        } else {
            mtm(Ji, Jj, T, 15, 15, 15);
            for (int a = 0; a < 225; a++) HB(i, i - j)[a] += T[a];
        }
        for (int b = 0; b < 15; b++) {
            double s1 = 0, s2 = 0;
            for (int a = 0; a < 15; a++) {
                s1 += Ji[a * 15 + b] * r[a];
                s2 += Jj[a * 15 + b] * r[a];
            }
            g[i * 15 + b] += s1;
            g[j * 15 + b] += s2;
        }
    }
    free(rJ);
    for (int f = 0; f < p->n_btw; f++) {
        double r[6], Ja[36], Jb[36], T[36];
        int a = p->btw_a[f], b = p->btw_b[f];
        vfo_between_factor(p->btw_data + (size_t)f * VFO_BTW_DATA, p->states + 16 * a,
                           p->states + 16 * b, 1, r, Ja, Jb);
        for (int i = 0; i < 6; i++) cost += 0.5 * r[i] * r[i];
        mtm(Ja, Ja, T, 6, 6, 6);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) HB(a, 0)[i * 15 + j] += T[i * 6 + j];
        mtm(Jb, Jb, T, 6, 6, 6);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) HB(b, 0)[i * 15 + j] += T[i * 6 + j];
        if (b > a) {
            mtm(Jb, Ja, T, 6, 6, 6);
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) HB(b, b - a)[i * 15 + j] += T[i * 6 + j];
        } else {
            mtm(Ja, Jb, T, 6, 6, 6);
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) HB(a, a - b)[i * 15 + j] += T[i * 6 + j];
        }
        for (int j = 0; j < 6; j++) {
            double s1 = 0, s2 = 0;
            for (int i = 0; i < 6; i++) {
                s1 += Ja[i * 6 + j] * r[i];
                s2 += Jb[i * 6 + j] * r[i];
            }
            g[a * 15 + j] += s1;
            g[b * 15 + j] += s2;
        }
    }
    for (int f = 0; f < p->n_prior; f++) {
        double r[15], J[225], T[225];
        int k = p->prior_k[f];
        vfo_prior_factor(p->prior_data + (size_t)f * VFO_PRIOR_DATA, p->states + 16 * k, r, J);
        for (int i = 0; i < 15; i++) cost += 0.5 * r[i] * r[i];
        mtm(J, J, T, 15, 15, 15);
        for (int i = 0; i < 225; i++) HB(k, 0)[i] += T[i];
        for (int j = 0; j < 15; j++) {
            double s = 0;
            for (int i = 0; i < 15; i++) s += J[i * 15 + j] * r[i];
            g[k * 15 + j] += s;
        }
    }
    if (p->marg && p->marg->on) {
        const vfo_marg* m = p->marg;
        double gm[27];
        cost += marg_cost(m, p->states, gm);
        for (int i = 0; i < 27; i++) {
            int ki, di;
            marg_index(i, &ki, &di);
            g[(m->k0 + ki) * 15 + di] += gm[i];
            for (int j = 0; j < 27; j++) {
                int kj, dj;
                marg_index(j, &kj, &dj);
                if (kj > ki) continue;               /* lower block triangle only */
                HB(m->k0 + ki, ki - kj)[di * 15 + dj] += m->L[i * 27 + j];
            }
        }
    }
    return cost;
}

/* Gauge floor of a marginal prior (vf_engine_opts.gauge_floor; k_marginalize does the same arithmetic).  Every factor of the
 * window is invariant under a global translation and a rotation about gravity: what the window knows about those four
 * directions is G^T L G of its marginal prior alone -- the memory of the anchor prior, which decays with every
 * marginalisation (measured: 2e-3, 1e-4, 1e-5, 8e-7 after 100, 500, 1000, 2000 updates of a 200-keyframe window) until it is
 * below the rounding of the 1e9-scale entries beside it; from there H is indefinite in float64, trials are rejected at
 * random and after ~3 000 updates solves fail.  The floor lifts the eigenvalues of G^T L G that have fallen below `floor` back
 * to it: information 1e-3 = a 30 m sigma on WHERE the window is, which constrains nothing the factors can see.  Prior
 * means are untouched (the term has zero gradient at the linearisation point, d = 0). */
static void jacobi4(double M[16], double V[16]) {
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 10; sweep++)
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                const double apq = M[p * 4 + q];
                if (apq == 0.0) continue;
                const double theta = (M[q * 4 + q] - M[p * 4 + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 4; k++) {          /* columns p, q of M and of V */
                    const double mkp = M[k * 4 + p], mkq = M[k * 4 + q];
                    M[k * 4 + p] = c * mkp - sn * mkq;
                    M[k * 4 + q] = sn * mkp + c * mkq;
                    const double vkp = V[k * 4 + p], vkq = V[k * 4 + q];
                    V[k * 4 + p] = c * vkp - sn * vkq;
                    V[k * 4 + q] = sn * vkp + c * vkq;
                }
                for (int k = 0; k < 4; k++) {          /* rows p, q of M */
                    const double mpk = M[p * 4 + k], mqk = M[q * 4 + k];
                    M[p * 4 + k] = c * mpk - sn * mqk;
                    M[q * 4 + k] = sn * mpk + c * mqk;
                }
            }
}
static void gauge_floor_apply(vfo_marg* mg, const double gravity[3], double floor) {
    if (!(floor > 0.0)) return;
    double G[27 * 4];
    memset(G, 0, sizeof(G));
    double gn = sqrt(dot3(gravity, gravity)), ez[3] = {0, 0, 1};
    if (gn > 0.0) for (int i = 0; i < 3; i++) ez[i] = -gravity[i] / gn;       /* "up" */
    const double* x0 = mg->xbar;
    for (int j = 0; j < 3; j++) {
        const double* x = mg->xbar + 16 * j;
        double R[9], dt[3] = {x[4] - x0[4], x[5] - x0[5], x[6] - x0[6]}, lever[3];
        vfo_quat_to_rot(x, R);
        cross3(ez, dt, lever);
        const int o = j == 0 ? 0 : 15 + 6 * (j - 1);
        for (int c = 0; c < 3; c++) {
            for (int a = 0; a < 3; a++) G[(o + 3 + c) * 4 + a] = R[a * 3 + c];                 /* translation along world axis a: R^T e_a */
            G[(o + c) * 4 + 3] = R[0 * 3 + c] * ez[0] + R[1 * 3 + c] * ez[1] + R[2 * 3 + c] * ez[2];   /* yaw: R^T e_z */
            G[(o + 3 + c) * 4 + 3] = R[0 * 3 + c] * lever[0] + R[1 * 3 + c] * lever[1] + R[2 * 3 + c] * lever[2];
        }
        if (j == 0) {
            double vy[3];
            cross3(ez, x + 7, vy);                                                            /* world-frame velocity turns with the yaw */
            for (int c = 0; c < 3; c++) G[(6 + c) * 4 + 3] = vy[c];
        }
    }
    for (int a = 0; a < 4; a++) {                       /* modified Gram-Schmidt */
        for (int b = 0; b < a; b++) {
            double s = 0.0;
            for (int i = 0; i < 27; i++) s += G[i * 4 + a] * G[i * 4 + b];
            for (int i = 0; i < 27; i++) G[i * 4 + a] -= s * G[i * 4 + b];
        }
        double nn = 0.0;
        for (int i = 0; i < 27; i++) nn += G[i * 4 + a] * G[i * 4 + a];
        nn = sqrt(nn);
        for (int i = 0; i < 27; i++) G[i * 4 + a] /= nn;
    }
    double T[27 * 4], M[16], V[16];
    for (int i = 0; i < 27; i++)
        for (int a = 0; a < 4; a++) {
            double s = 0.0;
            for (int j = 0; j < 27; j++) s += 0.5 * (mg->L[i * 27 + j] + mg->L[j * 27 + i]) * G[j * 4 + a];
            T[i * 4 + a] = s;
        }
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
            double s = 0.0;
            for (int i = 0; i < 27; i++) s += G[i * 4 + a] * T[i * 4 + b];
            M[a * 4 + b] = s;
        }
    for (int a = 0; a < 4; a++)
        for (int b = a + 1; b < 4; b++) M[a * 4 + b] = M[b * 4 + a] = 0.5 * (M[a * 4 + b] + M[b * 4 + a]);
    jacobi4(M, V);
    for (int e = 0; e < 4; e++) {
        const double lift = floor - M[e * 4 + e];
        if (!(lift > 0.0)) continue;
        double q[27];
        for (int i = 0; i < 27; i++) {
            double s = 0.0;
            for (int a = 0; a < 4; a++) s += G[i * 4 + a] * V[a * 4 + e];
            q[i] = s;
        }
        for (int i = 0; i < 27; i++)
            for (int j = 0; j < 27; j++) mg->L[i * 27 + j] += lift * q[i] * q[j];
    }
}

int vfo_marginalize(const vfo_problem* p, int m, vfo_marg* out) { return vfo_marginalize_floor(p, m, 0.0, out); }

int vfo_marginalize_floor(const vfo_problem* p, int m, double gauge_floor, vfo_marg* out) {
    /* variables [m:15][m+1:15][m+2 pose:6][m+3 pose:6] = 42; only factors touching m */
    double A[42 * 42], b[42];
    memset(A, 0, sizeof(A));
    memset(b, 0, sizeof(b));
    static const int OFF[4] = {0, 15, 30, 36};
    for (int f = 0; f < p->n_imu; f++) {
        if (p->imu_i[f] != m) continue;
        if (p->imu_j[f] != m + 1) return -1;
        double r[15], J[450];
        vfo_imu_factor(p->imu_data + (size_t)f * VFO_IMU_DATA, p->gravity, p->states + 16 * m,
                       p->states + 16 * (m + 1), 1, r, J);
        for (int a = 0; a < 30; a++) {
            const int ia = a < 15 ? IMU_COL_I[a] : IMU_COL_J[a - 15];
            for (int r0 = 0; r0 < 15; r0++) b[a] += J[r0 * 30 + ia] * r[r0];
            for (int c = 0; c < 30; c++) {
                const int ic = c < 15 ? IMU_COL_I[c] : IMU_COL_J[c - 15];
                double s = 0;
                for (int r0 = 0; r0 < 15; r0++) s += J[r0 * 30 + ia] * J[r0 * 30 + ic];
                A[a * 42 + c] += s;
            }
        }
    }
    for (int f = 0; f < p->n_btw; f++) {
        if (p->btw_a[f] != m) continue;
        const int d = p->btw_b[f] - m;
        if (d < 1 || d > 3) return -1;
        double r[6], Ja[36], Jb[36];
        vfo_between_factor(p->btw_data + (size_t)f * VFO_BTW_DATA, p->states + 16 * m,
                           p->states + 16 * (m + d), 1, r, Ja, Jb);
        const int ob = OFF[d];
        for (int a = 0; a < 6; a++) {
            for (int r0 = 0; r0 < 6; r0++) { b[a] += Ja[r0 * 6 + a] * r[r0]; b[ob + a] += Jb[r0 * 6 + a] * r[r0]; }
            for (int c = 0; c < 6; c++) {
                double saa = 0, sab = 0, sbb = 0;
                for (int r0 = 0; r0 < 6; r0++) {
                    saa += Ja[r0 * 6 + a] * Ja[r0 * 6 + c];
                    sab += Jb[r0 * 6 + a] * Ja[r0 * 6 + c];
                    sbb += Jb[r0 * 6 + a] * Jb[r0 * 6 + c];
                }
                A[a * 42 + c] += saa;
                A[(ob + a) * 42 + c] += sab;
                A[c * 42 + ob + a] += sab;
                A[(ob + a) * 42 + ob + c] += sbb;
            }
        }
    }
    for (int f = 0; f < p->n_prior; f++) {
        if (p->prior_k[f] != m) continue;
        double r[15], J[225];
        vfo_prior_factor(p->prior_data + (size_t)f * VFO_PRIOR_DATA, p->states + 16 * m, r, J);
        for (int a = 0; a < 15; a++) {
            for (int r0 = 0; r0 < 15; r0++) b[a] += J[r0 * 15 + a] * r[r0];
            for (int c = 0; c < 15; c++) {
                double s = 0;
                for (int r0 = 0; r0 < 15; r0++) s += J[r0 * 15 + a] * J[r0 * 15 + c];
                A[a * 42 + c] += s;
            }
        }
    }
    if (p->marg && p->marg->on) {
        if (p->marg->k0 != m) return -1;
        static const int MAP[27] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                    30, 31, 32, 33, 34, 35};
        double gm[27];
        marg_cost(p->marg, p->states, gm);
        for (int i = 0; i < 27; i++) {
            b[MAP[i]] += gm[i];
            for (int j = 0; j < 27; j++) A[MAP[i] * 42 + MAP[j]] += p->marg->L[i * 27 + j];
        }
    }
    /* Cholesky of the 15x15 pivot block, then Schur complement onto the remaining 27 */
    double Lm[225];
    memset(Lm, 0, sizeof(Lm));
    for (int j = 0; j < 15; j++) {
        double s = A[j * 42 + j];
        for (int k = 0; k < j; k++) s -= Lm[j * 15 + k] * Lm[j * 15 + k];
        if (!(s > 0.0)) return -1;
        Lm[j * 15 + j] = sqrt(s);
        for (int i = j + 1; i < 15; i++) {
            double v = A[i * 42 + j];
            for (int k = 0; k < j; k++) v -= Lm[i * 15 + k] * Lm[j * 15 + k];
            Lm[i * 15 + j] = v / Lm[j * 15 + j];
        }
    }
    double Y[27 * 15], y1[15];   /* Y = A21 L^-T, y1 = L^-1 b1 */
    for (int r0 = 0; r0 < 27; r0++)
        for (int c = 0; c < 15; c++) {
            double v = A[(15 + r0) * 42 + c];
            for (int k = 0; k < c; k++) v -= Y[r0 * 15 + k] * Lm[c * 15 + k];
            Y[r0 * 15 + c] = v / Lm[c * 15 + c];
        }
    for (int c = 0; c < 15; c++) {
        double v = b[c];
        for (int k = 0; k < c; k++) v -= Lm[c * 15 + k] * y1[k];
        y1[c] = v / Lm[c * 15 + c];
    }
    memset(out, 0, sizeof(*out));
    out->on = 1;
    out->k0 = m + 1;
    for (int i = 0; i < 27; i++) {
        double e = b[15 + i];
        for (int k = 0; k < 15; k++) e -= Y[i * 15 + k] * y1[k];
        out->eta[i] = e;
        for (int j = 0; j < 27; j++) {
            double s = A[(15 + i) * 42 + 15 + j];
            for (int k = 0; k < 15; k++) s -= Y[i * 15 + k] * Y[j * 15 + k];
            out->L[i * 27 + j] = s;
        }
    }
    /* the 27 kept dofs are [m+1: 15][m+2 pose][m+3 pose]: exactly rows 15..41 of the 42-vector */
    memcpy(out->xbar, p->states + 16 * (m + 1), sizeof(double) * 48);
    for (int i = 0; i < 27; i++)             /* (the device stores the symmetric part) */
        for (int j = i + 1; j < 27; j++) out->L[i * 27 + j] = out->L[j * 27 + i] = 0.5 * (out->L[i * 27 + j] + out->L[j * 27 + i]);
    gauge_floor_apply(out, p->gravity, gauge_floor);
    return 0;
}

/* workspace form: L = n_kf*15 x ((w+1)*15) doubles, y = n_kf*15 doubles, owned by the caller (vfo_lm allocates them
 * once per solve instead of once per trial) */
static int band_solve_ws(int n_kf, int w, const double* Hband, const double* g, double lambda,
                         double* delta, double* L, double* y) {
    /* scalar banded Cholesky of (H + lambda I), then delta = -(L L^T)^{-1} g */
    const int n = n_kf * 15, bw = (w + 1) * 15 - 1, ld = bw + 1;
    memset(L, 0, sizeof(double) * (size_t)n * ld);
#define LB(i, j) L[(size_t)(i) * ld + ((j) - (i) + bw)]
    for (int k = 0; k < n_kf; k++)
        for (int d = 0; d <= w && d <= k; d++) {
            const double* B = HB(k, d);
            for (int a = 0; a < 15; a++)
                for (int b = 0; b < 15; b++) {
                    int i = k * 15 + a, j = (k - d) * 15 + b;
                    if (j <= i) LB(i, j) = B[a * 15 + b];
                }
        }
    for (int i = 0; i < n; i++) LB(i, i) += lambda;
    int rc = 0;
    for (int j = 0; j < n && rc == 0; j++) {
        int k0 = j - bw > 0 ? j - bw : 0;
        double s = LB(j, j);
        for (int k = k0; k < j; k++) s -= LB(j, k) * LB(j, k);
        if (!(s > 0.0)) { rc = -1; break; }
        double ljj = sqrt(s);
        LB(j, j) = ljj;
        int i1 = j + bw < n - 1 ? j + bw : n - 1;
        for (int i = j + 1; i <= i1; i++) {
            int ki = i - bw > 0 ? i - bw : 0;
            double v = LB(i, j);
            for (int k = ki; k < j; k++) v -= LB(i, k) * LB(j, k);
            LB(i, j) = v / ljj;
        }
    }
    if (rc == 0) {
        for (int i = 0; i < n; i++) {
            int k0 = i - bw > 0 ? i - bw : 0;
            double s = -g[i];
            for (int k = k0; k < i; k++) s -= LB(i, k) * y[k];
            y[i] = s / LB(i, i);
        }
        for (int i = n - 1; i >= 0; i--) {
            int k1 = i + bw < n - 1 ? i + bw : n - 1;
            double s = y[i];
            for (int k = i + 1; k <= k1; k++) s -= LB(k, i) * delta[k];
            delta[i] = s / LB(i, i);
        }
    }
#undef LB
    return rc;
}

int vfo_band_solve(int n_kf, int w, const double* Hband, const double* g, double lambda,
                   double* delta) {
    const size_t n = (size_t)n_kf * 15, ld = (size_t)(w + 1) * 15;
    double* L = (double*)malloc(sizeof(double) * n * ld);
    double* y = (double*)malloc(sizeof(double) * n);
    const int rc = band_solve_ws(n_kf, w, Hband, g, lambda, delta, L, y);
    free(L);
    free(y);
    return rc;
}

/* ---- refined solve (the restatement of csrc/vf_refine.hip, formula for formula): conjugate gradients on
 * (J^T J + lambda I) d = -J^T r with the operator applied THROUGH the stored Jacobians -- never through H -- and the band
 * Cholesky solve as preconditioner.  The reference itself factorises by QR (GraphManager.cpp:38); normal equations of a long
 * chain of combined-IMU factors lose its softest modes to rounding (cond ~ n^4), the Jacobian does not. */
typedef struct {
    double *imu;   /* per IMU factor: r(15) Ji(15x15) Jj(15x15), columns in per-keyframe tangent order */
    double *btw;   /* per between factor: r(6) Ja(36) Jb(36) */
    double *pri;   /* per prior: r(15) J(225) */
} lin_store;

static void lin_build(const vfo_problem* p, lin_store* S) {
    S->imu = (double*)malloc(sizeof(double) * (size_t)(p->n_imu > 0 ? p->n_imu : 1) * 465);
    S->btw = (double*)malloc(sizeof(double) * (size_t)(p->n_btw > 0 ? p->n_btw : 1) * 78);
    S->pri = (double*)malloc(sizeof(double) * (size_t)(p->n_prior > 0 ? p->n_prior : 1) * 240);
    for (int f = 0; f < p->n_imu; f++) {
        double r[15], J[450];
        double* o = S->imu + (size_t)f * 465;
        vfo_imu_factor(p->imu_data + (size_t)f * VFO_IMU_DATA, p->gravity, p->states + 16 * p->imu_i[f],
                       p->states + 16 * p->imu_j[f], 1, r, J);
        memcpy(o, r, sizeof(r));
        for (int a = 0; a < 15; a++)
            for (int b = 0; b < 15; b++) {
                o[15 + a * 15 + b] = J[a * 30 + IMU_COL_I[b]];
                o[240 + a * 15 + b] = J[a * 30 + IMU_COL_J[b]];
            }
    }
    for (int f = 0; f < p->n_btw; f++) {
        double* o = S->btw + (size_t)f * 78;
        vfo_between_factor(p->btw_data + (size_t)f * VFO_BTW_DATA, p->states + 16 * p->btw_a[f],
                           p->states + 16 * p->btw_b[f], 1, o, o + 6, o + 42);
    }
    for (int f = 0; f < p->n_prior; f++) {
        double* o = S->pri + (size_t)f * 240;
        vfo_prior_factor(p->prior_data + (size_t)f * VFO_PRIOR_DATA, p->states + 16 * p->prior_k[f], o, o + 15);
    }
}
static void lin_free(lin_store* S) { free(S->imu); free(S->btw); free(S->pri); }

/* out = J^T (J v) + lambda v  (+ the marginal prior's information times v: it is kept in information form) */
static void lin_apply(const vfo_problem* p, const lin_store* S, double lambda, const double* v, double* out) {
    const int n = p->n_kf;
    for (int i = 0; i < n * 15; i++) out[i] = lambda * v[i];
    for (int f = 0; f < p->n_imu; f++) {
        const double *Ji = S->imu + (size_t)f * 465 + 15, *Jj = Ji + 225;
        const double *vi = v + 15 * p->imu_i[f], *vj = v + 15 * p->imu_j[f];
        double u[15];
        for (int a = 0; a < 15; a++) {
            double s = 0.0;
            for (int b = 0; b < 15; b++) s += Ji[a * 15 + b] * vi[b] + Jj[a * 15 + b] * vj[b];
            u[a] = s;
        }
        double *oi = out + 15 * p->imu_i[f], *oj = out + 15 * p->imu_j[f];
        for (int b = 0; b < 15; b++) {
            double s1 = 0.0, s2 = 0.0;
            for (int a = 0; a < 15; a++) { s1 += Ji[a * 15 + b] * u[a]; s2 += Jj[a * 15 + b] * u[a]; }
            oi[b] += s1;
            oj[b] += s2;
        }
    }
    for (int f = 0; f < p->n_btw; f++) {
        const double *Ja = S->btw + (size_t)f * 78 + 6, *Jb = Ja + 36;
        const double *va = v + 15 * p->btw_a[f], *vb = v + 15 * p->btw_b[f];
        double u[6];
        for (int a = 0; a < 6; a++) {
            double s = 0.0;
            for (int b = 0; b < 6; b++) s += Ja[a * 6 + b] * va[b] + Jb[a * 6 + b] * vb[b];
            u[a] = s;
        }
        double *oa = out + 15 * p->btw_a[f], *ob = out + 15 * p->btw_b[f];
        for (int b = 0; b < 6; b++) {
            double s1 = 0.0, s2 = 0.0;
            for (int a = 0; a < 6; a++) { s1 += Ja[a * 6 + b] * u[a]; s2 += Jb[a * 6 + b] * u[a]; }
            oa[b] += s1;
            ob[b] += s2;
        }
    }
    for (int f = 0; f < p->n_prior; f++) {
        const double* J = S->pri + (size_t)f * 240 + 15;
        const double* vk = v + 15 * p->prior_k[f];
        double u[15];
        for (int a = 0; a < 15; a++) {
            double s = 0.0;
            for (int b = 0; b < 15; b++) s += J[a * 15 + b] * vk[b];
            u[a] = s;
        }
        double* ok = out + 15 * p->prior_k[f];
        for (int b = 0; b < 15; b++) {
            double s = 0.0;
            for (int a = 0; a < 15; a++) s += J[a * 15 + b] * u[a];
            ok[b] += s;
        }
    }
    if (p->marg && p->marg->on) {
        const vfo_marg* m = p->marg;
        for (int i = 0; i < 27; i++) {
            int ki, di;
            marg_index(i, &ki, &di);
            double s = 0.0;
            for (int j = 0; j < 27; j++) {
                int kj, dj;
                marg_index(j, &kj, &dj);
                s += m->L[i * 27 + j] * v[(m->k0 + kj) * 15 + dj];
            }
            out[(m->k0 + ki) * 15 + di] += s;
        }
    }
}

/* d (the plain normal-equation solution on entry) refined by at most `refine` corrections; L, y: the band solver's
 * workspace; returns the corrections applied */
static int refine_solution(const vfo_problem* p, int w, const double* H, const double* g, double lambda, int refine,
                           double rel_stop, double* d, double* L, double* y) {
    const int N = p->n_kf * 15;
    lin_store S;
    lin_build(p, &S);
    double* x = (double*)malloc(sizeof(double) * (size_t)N * 5);
    double *dir = x + N, *Ap = x + 2 * (size_t)N, *nres = x + 3 * (size_t)N, *z = x + 4 * (size_t)N;
    memcpy(x, d, sizeof(double) * (size_t)N);
    lin_apply(p, &S, lambda, x, Ap);
    for (int i = 0; i < N; i++) nres[i] = g[i] + Ap[i];
    double rz = -1.0, rz0 = 0.0;
    int done = 0;
    for (int it = 0; it < refine; it++) {
        if (band_solve_ws(p->n_kf, w, H, nres, lambda, z, L, y) != 0) break;     /* z = M^-1 res */
        double rzn = 0.0;
        for (int i = 0; i < N; i++) rzn += -nres[i] * z[i];
        const int first = rz < 0.0;
        if (!(rzn > 0.0) || (!first && rzn <= rel_stop * rel_stop * rz0)) break;
        const double beta = first ? 0.0 : rzn / rz;
        for (int i = 0; i < N; i++) dir[i] = first ? z[i] : z[i] + beta * dir[i];
        rz = rzn;
        if (first) rz0 = rzn;
        lin_apply(p, &S, lambda, dir, Ap);
        double pAp = 0.0;
        for (int i = 0; i < N; i++) pAp += dir[i] * Ap[i];
        if (!(pAp > 0.0)) break;
        const double alpha = rz / pAp;
        for (int i = 0; i < N; i++) { x[i] += alpha * dir[i]; nres[i] += alpha * Ap[i]; }
        done++;
    }
    memcpy(d, x, sizeof(double) * (size_t)N);
    free(x);
    lin_free(&S);
    return done;
}

/* One reference-compat update on a batch problem: undamped Gauss-Newton at the current states (what one ISAM2::update
 * amounts to when every variable is relinearised, GraphManager.cpp:126-127), states <- states (+) delta.  refine > 0: the
 * step is refined as above.  Returns 0, or -1 when the normal equations are not positive definite. */
int vfo_gn_step(vfo_problem* p, int refine, double rel_stop, double* cost_before, int* corrections) {
    const int n = p->n_kf, w = vfo_bandwidth(p);
    double* H = (double*)malloc(sizeof(double) * (size_t)n * (w + 1) * 225);
    double* g = (double*)malloc(sizeof(double) * (size_t)n * 15);
    double* d = (double*)malloc(sizeof(double) * (size_t)n * 15);
    double* Lws = (double*)malloc(sizeof(double) * (size_t)n * 15 * (size_t)(w + 1) * 15);
    double* yws = (double*)malloc(sizeof(double) * (size_t)n * 15);
    const double c = vfo_assemble(p, w, H, g, 1);
    if (cost_before) *cost_before = c;
    int rc = band_solve_ws(n, w, H, g, 0.0, d, Lws, yws);
    if (rc == 0) {
        const int done = refine > 0 ? refine_solution(p, w, H, g, 0.0, refine, rel_stop, d, Lws, yws) : 0;
        if (corrections) *corrections = done;
        double x[16];
        for (int k = 0; k < n; k++) {
            vfo_retract(p->states + 16 * k, d + 15 * k, x);
            memcpy(p->states + 16 * k, x, sizeof(x));
        }
    }
    free(H); free(g); free(d); free(Lws); free(yws);
    return rc;
}

double vfo_lm(vfo_problem* p, const vfo_lm_opts* o, double* costs_out, int* accepted_out) {
    /* One trial per iteration: linearise at x, solve (H + lambda I) d = -g, accept iff the
     * cost decreases (lambda /= down) else reject (lambda *= up).  Defaults follow
     * gtsam::LevenbergMarquardtParams (lambdaInitial 1e-5, lambdaFactor 10), the optimiser the
     * reference leaves commented out at GraphManager.cpp:128-129.
     * o->excursion = W > 0: the non-monotone rule of the engine (include/vilfusion.h "lm_excursion"; k_decide): up to W
     * consecutive cost-raising trials are kept provisionally (accepted_out = 2), judged against the cost of the point the
     * excursion left; the W+1-th that still is not below it restores that point (accepted_out = 3). */
    const int n = p->n_kf, w = vfo_bandwidth(p);
    double* H = (double*)malloc(sizeof(double) * (size_t)n * (w + 1) * 225);
    double* g = (double*)malloc(sizeof(double) * (size_t)n * 15);
    double* d = (double*)malloc(sizeof(double) * (size_t)n * 15);
    double* xs = (double*)malloc(sizeof(double) * (size_t)n * 16);
    double* xbest = (double*)malloc(sizeof(double) * (size_t)n * 16);
    /* buffers of a trial, allocated once per solve: the trial's normal equations and the band solver's workspace */
    double* Hn = (double*)malloc(sizeof(double) * (size_t)n * (w + 1) * 225);
    double* gn = (double*)malloc(sizeof(double) * (size_t)n * 15);
    double* Lws = (double*)malloc(sizeof(double) * (size_t)n * 15 * (size_t)(w + 1) * 15);
    double* yws = (double*)malloc(sizeof(double) * (size_t)n * 15);
    double lambda = o->lambda0;
    double cost = vfo_assemble(p, w, H, g, o->n_threads);
    if (costs_out) costs_out[0] = cost;
    int converged = 0, prov = 0;
    double ref_cost = cost;
    for (int it = 0; it < o->iterations; it++) {
        if (converged) {
            if (costs_out) costs_out[it + 1] = cost;
            if (accepted_out) accepted_out[it] = -1;
            continue;
        }
        int ok = band_solve_ws(n, w, H, g, lambda, d, Lws, yws) == 0, outcome = 0;
        const double refc = prov > 0 ? ref_cost : cost;
        if (ok) {
            if (o->refine > 0) refine_solution(p, w, H, g, lambda, o->refine, o->refine_rel_stop, d, Lws, yws);
            memcpy(xs, p->states, sizeof(double) * (size_t)n * 16);
            for (int k = 0; k < n; k++) vfo_retract(xs + 16 * k, d + 15 * k, p->states + 16 * k);
            double cn = vfo_assemble(p, w, Hn, gn, o->n_threads);
            int good = cn < refc + o->accept_rel * refc;
            if (o->min_model_fidelity > 0.0) {
                /* GTSAM's test: actual over predicted decrease, the prediction by the linearised (undamped) problem,
                 * 0.5 |r|^2 - 0.5 |r + J d|^2 = -g.d - 0.5 d^T H d, which with (H + lambda I) d = -g is 0.5 (lambda |d|^2 - g.d) */
                double gd = 0.0, dd = 0.0;
                for (int i = 0; i < n * 15; i++) { gd += g[i] * d[i]; dd += d[i] * d[i]; }
                const double predicted = 0.5 * (lambda * dd - gd);
                good = predicted > 0.0 && (refc - cn) > o->min_model_fidelity * predicted;
            }
            outcome = good ? 1 : (o->excursion > 0 && prov < o->excursion) ? 2 : (prov > 0 ? 3 : 0);
            /* termination (gtsam checkConvergence), also on a rejected trial within the tolerance: the
             * window then sits at its rounding floor */
            if (outcome < 2 && (o->abs_tol > 0.0 || o->rel_tol > 0.0) &&
                (fabs(refc - cn) <= o->abs_tol || fabs(refc - cn) <= o->rel_tol * refc)) converged = 1;
            if (outcome == 1 || outcome == 2) { /* NaN compares false -> not 1 */
                if (outcome == 2 && prov == 0) {
                    ref_cost = cost;
                    memcpy(xbest, xs, sizeof(double) * (size_t)n * 16);
                }
                cost = cn;
                double* t = H; H = Hn; Hn = t;      /* the trial's normal equations become the current ones */
                t = g; g = gn; gn = t;
                prov = outcome == 1 ? 0 : prov + 1;
            } else {
                memcpy(p->states, xs, sizeof(double) * (size_t)n * 16);
            }
        } else if (prov > 0) outcome = 3;
        if (outcome == 3) {           /* the excursion failed: back to the point it left */
            memcpy(p->states, xbest, sizeof(double) * (size_t)n * 16);
            cost = vfo_assemble(p, w, H, g, o->n_threads);
            prov = 0;
        }
        if (outcome == 1) {
            lambda /= o->lambda_down;
            if (lambda < o->lambda_min) lambda = o->lambda_min;
        } else if (outcome == 2) {
            lambda /= o->lambda_down * o->lambda_down;
            if (lambda < o->lambda_min) lambda = o->lambda_min;
        } else {
            lambda *= o->lambda_up;
            if (lambda > o->lambda_max) lambda = o->lambda_max;
        }
        if (costs_out) costs_out[it + 1] = cost;
        if (accepted_out) accepted_out[it] = outcome;
    }
    if (prov > 0) {                   /* an excursion still open when the trials run out is undone */
        memcpy(p->states, xbest, sizeof(double) * (size_t)n * 16);
        if (costs_out) costs_out[o->iterations] = ref_cost;
    }
    free(H); free(g); free(d); free(xs); free(xbest);
    free(Hn); free(gn); free(Lws); free(yws);
    return lambda;
}
