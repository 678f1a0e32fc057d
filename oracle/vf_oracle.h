/*
 * vf_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C float64 restatement of the arithmetic behind the reference's hot path
 * (gtsam_fusion::GraphManager::solve and what it calls into GTSAM for).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product library (vil_sensor_fusion_amd/csrc) never links or calls it.
 *
 * PARITY STATUS
 *   pinned   : IMU preintegration mean + IMUManager::getFactor interpolation rule
 *              (KAT gtsam_fusion/test/UnitTests.cpp:58-66), poseDiff / key numbering
 *              (UnitTests.cpp:200-233), SO(3)/SE(3) exp/log (scipy expm/logm),
 *              every Jacobian (central differences against the stated retraction).
 *   UNPINNED : whitened residual/Jacobian VALUES, the 15x15 preintegrated covariance,
 *              and the LM trajectory versus GTSAM itself.  GTSAM (un-pinned version,
 *              4.0.3 <= v < 4.3 by API evidence) is not in /root/reference and is not
 *              installable here; its published algorithm is restated from memory of
 *              the public sources and every convention choice is listed in DESIGN.md.
 *
 * Conventions: quaternions (w,x,y,z); matrices row-major; Pose3 tangent [omega, v];
 * per-keyframe state = q(4) t(3) v(3) ba(3) bg(3) = 16 doubles; per-keyframe tangent
 * [dtheta(3) dp(3) dv(3) dba(3) dbg(3)] = 15.
 */
#ifndef VF_ORACLE_H
#define VF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFO_STATE_DIM 16
#define VFO_TANGENT_DIM 15
#define VFO_IMU_DATA 190   /* dt(1) delta(9) bhat(6) H(54) R_upper(120) */
#define VFO_BTW_DATA 28    /* q(4) t(3) R_upper(21) */
#define VFO_PRIOR_DATA 31  /* mean state(16) sigma(15) */

/* PreintegratedCombinedMeasurements::Params as configured by
 * gtsam_fusion/src/gtsam_fusion/ImuManagerRos.cpp:14-36 (isotropic covariances,
 * MakeSharedU => n_gravity = (0,0,-9.81)). */
typedef struct {
    double acc_cov;            /* setAccelerometerCovariance  (ident * accel)       */
    double gyro_cov;           /* setGyroscopeCovariance      (ident * gyro)        */
    double int_cov;            /* setIntegrationCovariance    (ident * integration) */
    double bias_acc_cov;       /* setBiasAccCovariance        (ident * biasAcc)     */
    double bias_omega_cov;     /* setBiasOmegaCovariance      (ident * biasOmega)   */
    double bias_acc_omega_int; /* setBiasAccOmegaInt (Matrix6::Identity * biasAccInt) */
    double gravity[3];         /* n_gravity */
} vfo_imu_params;

/* State of a PreintegratedCombinedMeasurements object (tangent preintegration). */
typedef struct {
    double dt;        /* deltaTij */
    double d[9];      /* preintegrated_: theta, position, velocity */
    double bhat[6];   /* biasHat_: acc, gyro */
    double H[54];     /* 9x6 row-major [preintegrated_H_biasAcc_ | preintegrated_H_biasOmega_] */
    double cov[225];  /* preintMeasCov_ 15x15, order [theta pos vel biasAcc biasOmega] */
} vfo_pim;

/* ---- Lie-group primitives (exported for the tests) ---- */
void vfo_so3_exp(const double w[3], double R[9]);
void vfo_so3_log(const double R[9], double w[3]);
void vfo_so3_jr(const double w[3], double J[9]);       /* Rot3::ExpmapDerivative  */
void vfo_so3_jr_inv(const double w[3], double J[9]);   /* Rot3::LogmapDerivative  */
void vfo_quat_to_rot(const double q[4], double R[9]);
void vfo_rot_to_quat(const double R[9], double q[4]);
void vfo_se3_exp(const double xi[6], double R[9], double t[3]);      /* Pose3::Expmap */
void vfo_se3_log(const double R[9], const double t[3], double xi[6]);/* Pose3::Logmap */
void vfo_se3_jr_inv(const double xi[6], double J[36]);               /* Pose3::LogmapDerivative */

/* ---- IMU preintegration (IMUManager.cpp:27-74 + GTSAM PIM) ---- */
void vfo_pim_reset(vfo_pim* p, const double bhat[6]);
void vfo_pim_integrate(vfo_pim* p, const vfo_imu_params* prm, const double acc[3],
                       const double gyro[3], double dt);
/* IMUManager::getFactor(start,end,...) on a time-sorted sample buffer
 * (t[n], acc[3n], gyro[3n]).  *head is the deque front on entry/exit. */
int vfo_imu_get_factor(const double* t, const double* acc, const double* gyro, int n, int* head,
                       double start, double end, const double bias[6], const vfo_imu_params* prm,
                       vfo_pim* out);
/* R upper-triangular with R^T R = cov^{-1} (noiseModel::Gaussian::Covariance).
 * Returns 0, or -1 when cov is not SPD. Rp = packed upper, row-major, n(n+1)/2. */
int vfo_sqrt_info_upper(const double* cov, int n, double* Rp);
/* pack a pim + its noise model into the 190-double factor record */
int vfo_pim_to_record(const vfo_pim* p, double rec[VFO_IMU_DATA]);
/* PreintegrationBase::predict (GraphManager.cpp:153) */
void vfo_predict(const double rec[VFO_IMU_DATA], const double gravity[3],
                 const double state_i[16], double state_j[16]);

/* ---- factor residuals + Jacobians ---- */
/* CombinedImuFactor::evaluateError. J = 15x30 row-major, column order
 * [X_i(6) V_i(3) X_j(6) V_j(3) B_i(6) B_j(6)] (IMUManager.cpp:68-73 key order).
 * whiten != 0 applies R. */
void vfo_imu_factor(const double rec[VFO_IMU_DATA], const double gravity[3], const double xi[16],
                    const double xj[16], int whiten, double r[15], double J[450]);
/* BetweenFactor<Pose3>::evaluateError. Ja, Jb 6x6 row-major. */
void vfo_between_factor(const double rec[VFO_BTW_DATA], const double xa[16], const double xb[16],
                        int whiten, double r[6], double Ja[36], double Jb[36]);
/* the three PriorFactors of GraphManager.cpp:33-35 as one 15-row diagonal factor */
void vfo_prior_factor(const double rec[VFO_PRIOR_DATA], const double x[16], double r[15],
                      double J[225]);
/* Retract of one keyframe: Pose3 full Expmap, vector add for v, bias. */
void vfo_retract(const double x[16], const double delta[15], double out[16]);

/* ---- window problem + LM ---- */
/* Marginal prior left by marginalising the keyframe in front of the window (fixed-lag, SURVEY
 * section 8f-3; no reference code -- the reference's iSAM2 never marginalises).  Gaussian on
 * d = [Local(xbar_0 -> x_k0) (15) ; pose part of Local(xbar_1 -> x_k0+1) (6) ; same for k0+2 (6)]:
 *   cost(d) = 0.5 d^T L d + eta^T d,  fixed linearisation point xbar, dLocal/dx taken as identity. */
typedef struct {
    int on;                      /* 0 = absent */
    int k0;                      /* first keyframe (window-local index) */
    double xbar[48];             /* linearisation states of k0, k0+1, k0+2 */
    double L[729];               /* 27x27 information, row-major */
    double eta[27];              /* gradient at d = 0 */
} vfo_marg;

typedef struct {
    int n_kf;
    double* states;              /* n_kf*16, in/out */
    int n_imu;  const int32_t* imu_i;  const int32_t* imu_j;  const double* imu_data;
    int n_btw;  const int32_t* btw_a;  const int32_t* btw_b;  const double* btw_data;
    int n_prior; const int32_t* prior_k; const double* prior_data;
    double gravity[3];
    const vfo_marg* marg;        /* optional marginal prior (NULL = none) */
} vfo_problem;

typedef struct {
    double lambda0, lambda_up, lambda_down, lambda_min, lambda_max;
    int iterations;              /* fixed trip count (one trial per iteration) */
    int n_threads;               /* >1: OpenMP over factors in linearisation */
    double rel_tol, abs_tol;     /* > 0: stop after an accepted trial whose cost decrease is <= abs_tol or
                                    <= rel_tol * cost (gtsam LevenbergMarquardtOptimizer checkConvergence) */
    double accept_rel;           /* a trial is accepted iff  new cost < cost + accept_rel * cost  (0: strict decrease; the
                                    engine's default is 1e-9: the rounding floor of the cost sum) */
    int refine;                  /* > 0: every solve is followed by this many conjugate-gradient corrections through the
                                    Jacobians (vf_engine_opts.refine_iterations; csrc/vf_refine.hip) */
    double refine_rel_stop;      /* vf_engine_opts.refine_rel_stop */
    int excursion;               /* vf_engine_opts.lm_excursion: provisional cost-raising trials per excursion (0: classical) */
    double min_model_fidelity;   /* > 0: GTSAM's accept rule instead of accept_rel (gtsam::LevenbergMarquardtOptimizer::tryLambda with
                                    LevenbergMarquardtParams::minModelFidelity, 1e-3 there): a trial is accepted iff
                                    (cost - new cost) / (cost - cost of the LINEARISED problem at delta) > this;
                                    vf_engine_opts.min_model_fidelity */
} vfo_lm_opts;

/* total cost 0.5*sum |r|^2 at the current states */
double vfo_cost(const vfo_problem* p);
/* block half-bandwidth (in keyframes) implied by the factors */
int vfo_bandwidth(const vfo_problem* p);
/* Assemble banded normal equations: Hband[(n_kf) * (w+1) * 225] block rows
 * (block d of row k = H[k][k-d]), g[n_kf*15]. Returns cost. */
double vfo_assemble(const vfo_problem* p, int w, double* Hband, double* g, int n_threads);
/* Solve (H + lambda I) delta = -g by banded Cholesky. 0 ok / -1 not PD. */
int vfo_band_solve(int n_kf, int w, const double* Hband, const double* g, double lambda,
                   double* delta);
/* One undamped Gauss-Newton update at the current states (a reference-compat update with every variable relinearised,
 * GraphManager.cpp:126-127), refined by `refine` corrections when > 0; states <- states (+) delta.  0 ok / -1 not PD. */
int vfo_gn_step(vfo_problem* p, int refine, double rel_stop, double* cost_before, int* corrections);
/* Fixed-trip LM; costs_out[iterations+1] (cost after each iteration, [0] = initial),
 * accepted_out[iterations] (1 accepted, 0 rejected, -1 = trial not run: converged earlier; with excursions also 2 = kept
 * provisionally, 3 = excursion failed, its starting point restored). Returns final lambda. */
double vfo_lm(vfo_problem* p, const vfo_lm_opts* o, double* costs_out, int* accepted_out);

/* d (27) of a marginal prior at the current states */
void vfo_marg_delta(const vfo_marg* m, const double* states, double d[27]);
/* Marginalise keyframe m (window-local; must be the oldest one): Schur complement of all factors
 * touching m (its prior / marginal prior, imu factor m -> m+1, between factors starting at m),
 * linearised at the current states, onto [m+1: 15][m+2: pose][m+3: pose]. out->k0 = m+1. */
int vfo_marginalize(const vfo_problem* p, int m, vfo_marg* out);
/* ... with the gauge floor of vf_engine_opts.gauge_floor (0 = none): eigenvalues of the prior's information about the window's
 * global translation and yaw that have decayed below `gauge_floor` are lifted back to it */
int vfo_marginalize_floor(const vfo_problem* p, int m, double gauge_floor, vfo_marg* out);

#ifdef __cplusplus
}
#endif
#endif
