/*
 * vilfusion.h -- C ABI of libvilfusion.so, the MI355X (gfx950) drop-in for the arithmetic
 * behind gtsam_fusion's GraphManager / IMUManager.
 *
 * The reference has no FFI layer: the only seam on its hot path is the public C++ API of
 * VILFusion::GraphManager (gtsam_fusion/include/gtsam_fusion/GraphManager.h:40-96) and
 * VILFusion::IMUManager (gtsam_fusion/include/gtsam_fusion/IMUManager.h:22-27).  Each entry
 * point below cites the reference member it replaces.  All pointers are borrowed for the
 * duration of the call; outputs are caller-allocated; no call throws; every call returns an
 * int status (0 = VF_OK, <0 = error, message via vf_last_error()).  No torch / GTSAM types.
 *
 * Layout conventions (host side of the ABI, plain AoS, float64):
 *   quaternion (w,x,y,z);  state16 = q(4) t(3) v(3) bias_acc(3) bias_gyro(3);
 *   tangent15 = dtheta(3) dp(3) dv(3) dba(3) dbg(3);  Pose3 tangent order [rot, trans].
 *   imu record (190)     = dt, delta(9), bias_hat(6), H(9x6 row-major), R(120 packed upper)
 *   between record (28)  = q_meas(4), t_meas(3), R(21 packed upper), R^T R = cov^-1
 *   prior record (31)    = mean state16, sigma(15)
 * On the device these live as AoSoA tiles of 64 factors (DESIGN.md "Data layout in HBM").
 */
#ifndef VILFUSION_H
#define VILFUSION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VF_OK 0
#define VF_ERR_INVALID (-1)       /* bad argument */
#define VF_ERR_BAD_KEY (-2)       /* unknown / out-of-window key */
#define VF_ERR_NOT_SPD (-3)       /* covariance not symmetric positive definite */
#define VF_ERR_INDETERMINATE (-4) /* normal equations not positive definite */
#define VF_ERR_DEVICE (-5)        /* HIP runtime error */
#define VF_ERR_CAPACITY (-6)      /* window / slot capacity exceeded */
#define VF_ERR_NO_DEVICE (-7)     /* no gfx950 device visible: the library never falls back to CPU */

#define VF_STATE_DIM 16
#define VF_TANGENT_DIM 15
#define VF_IMU_RECORD 190
#define VF_BTW_RECORD 28
#define VF_PRIOR_RECORD 31
#define VF_MAX_BANDWIDTH 3
#define VF_MAX_EXTRA 8            /* "far" between factors per window (vf_engine_set_extra_between) unless the handle asks for more ... */
#define VF_MAX_FAR_LIMIT 32       /* ... through max_far_factors, up to this many */

const char* vf_last_error(void);
const char* vf_version(void);
int vf_device_count(int* count);

/* ===================================================================== engine (batch)
 * B independent windows of up to `capacity` keyframes on one GPU.  This is the form the
 * hot path takes on the device; the GraphManager handle below is an engine with B = 1. */
typedef struct vf_engine vf_engine;

typedef struct {
    uint32_t struct_size;   /* sizeof(vf_engine_opts) in the header the CALLER was compiled against; vf_engine_default_opts sets
                               it.  The struct only ever grows at its end: a caller built against an older, shorter header is
                               served with the defaults for the fields it does not know; a size the library does not know
                               (larger than its own, or 0) is refused with VF_ERR_INVALID */
    int windows;            /* B */
    int capacity;           /* keyframes per window, rounded up to a multiple of 64 */
    int bandwidth;          /* max |a-b| of a between factor, 1..VF_MAX_BANDWIDTH */
    int device;             /* HIP device ordinal */
    double gravity[3];      /* n_gravity; MakeSharedU => (0,0,-9.81) (ImuManagerRos.cpp:16) */
    double lambda0, lambda_up, lambda_down, lambda_min, lambda_max; /* LM damping schedule */
    int chunks;             /* K4 form.  0 = chosen from the batch size: up to 128 windows -> partitioned
                               solve (chunks joined by 27-dof separators, about sqrt(n) of them for an
                               n-keyframe window: one-window latency); more windows -> one sweep per window
                               (throughput).  1 = always sweeps.  P >= 2 = partitioned solve with P chunks. */
    int cold_start;          /* != 0: every vf_engine_iterate linearises all factors (no warm start); default 0 */
    double accept_rel;       /* an LM trial is accepted iff  new cost < cost + accept_rel * cost.  Default 1e-9: the rounding
                                floor of the cost of a 1000-pose window (a sum of ~30 000 squared whitened residuals, the IMU
                                ones scaled by 5e4) is ~1e-10 of its value; with a strict "decreases" test (accept_rel = 0) a
                                converged window rejects about half of its Newton steps on the last bits of that sum, its
                                soft modes stop converging, and two float64 implementations end 1e-7 ... 1e-6 m apart
                                instead of 1e-9 (DESIGN.md "Accept rule at the floor").  Must be >= 0.
                                NOTE this accept test is the build's own.  Of gtsam::LevenbergMarquardtParams only lambdaInitial
                                (1e-5), lambdaFactor (10) and the termination tolerances (vf_engine_set_convergence) are
                                followed; GTSAM accepts a step on modelFidelity = actual / predicted decrease >
                                minModelFidelity (1e-3), which this library does not evaluate.  Where the optimum is does not
                                depend on either rule: tests/test_gpu_vs_qr_twin.py holds the result to an independent QR
                                optimiser that uses the gain-ratio test. */
    /* Refined solve (DESIGN.md "Refined solve"; csrc/vf_refine.hip).  The reference factorises by QR (GraphManager.cpp:38); the
     * device forms normal equations, whose condition number grows with the FOURTH power of the window length (a chain of
     * combined-IMU factors is a double integrator): ~1e11 at 1000 keyframes, beyond 1e19 at 10 000, where a float64 Cholesky
     * factor has the softest modes wrong by orders of magnitude and Gauss-Newton / LM creep instead of converging.  With
     * refinement every solve is followed by N conjugate-gradient corrections on (J^T J + lambda I) d = -J^T r with the
     * operator applied THROUGH J (accurate to cond(J), not cond(J)^2) and the Cholesky solve as preconditioner: each
     * correction costs one more solve with the same factor shape plus two passes over the Jacobians, and removes one of the
     * handful of modes the factor gets wrong.  -1 (default) = up to 12 corrections once a window is longer than
     * refine_min_keyframes, none otherwise (the headline windows are untouched); 0 = never; N = always N (at most 64).
     * Refining engines run the two-kernel form (K3 + K4).  Far factors of a refined engine are rows of the refinement's operator
     * too, and the Woodbury solve (band factor + the far factors' columns, computed once per trial) is its preconditioner. */
    int refine_iterations;
    int refine_min_keyframes;  /* default 1536: Gauss-Newton by normal equations alone contracts by 0.025 per update at 1 250 keyframes, 0.1 at
                                  1 500, 0.3 at 2 000, 0.7 at 3 000 and creeps beyond */
    double refine_rel_stop;    /* a window stops correcting once res . M^-1 res has fallen to this, squared, times its first value
                                  (default 1e-8), or stops being positive.  In auto mode (refine_iterations = -1) the tolerance EASES IN:
                                  1e-3 for a window of refine_min_keyframes, tightening log-linearly to this value at twice that length
                                  (what the normal equations get wrong grows smoothly with the window; a switch from no correction to
                                  "until 1e-8" at one length made a solve four times dearer from 1 500 to 1 600 keyframes); its remaining correction solves are skipped on the
                                  device (no host synchronisation).  How many corrections that takes grows with the window: 4 at
                                  1 600 keyframes, 5 at 2 500, 7 at 4 000, 9 at 6 000, 12 at 10 000 (tools/refine_trace_sizes.py) */
    /* Non-monotone LM ("excursions").  On a long window the Gauss-Newton step moves the far end by metres through
     * rotation-coupled dynamics; the stiff IMU residuals (sigma 2e-5 m) see the second-order part of that move, so the cost
     * RISES after the step (13 -> 1700 on the 10 000-pose window) and falls below its start only after one or two more
     * steps have corrected it (-> 10.78).  A monotone accept test rejects every such step and damped steps creep along the
     * valley (profiles/r04_config4_soft_mode.log; gtsam's own LM rule does the same).  With lm_excursion = W > 0 up to W
     * consecutive cost-raising trials are kept provisionally, each dividing lambda by lambda_down twice; a later trial whose
     * cost is below the cost the excursion started from accepts them all; if the W+1-th still is not, the starting point
     * is restored, lambda multiplied by lambda_up, and the trials count as rejected.  An excursion still open when a solve's
     * trials run out is undone.  Under this rule lambda PERSISTS from one solve to the next (the first starts from lambda0):
     * lambda0 = 1e-5 is six orders above the soft eigenvalues of a window long enough to be refined, and a solve of five
     * trials that starts there each time spends them all coming down, or repeats the same failed excursion for ever.  0 = the classical rule; -1 (default) = 3 on engines that refine (windows longer than
     * refine_min_keyframes), 0 otherwise. */
    int lm_excursion;
    /* Gauge floor of the fixed-lag marginal prior (vf_engine_marginalize / vf_engine_slide(marginalize = 1); no reference code:
     * the reference's iSAM2 graph never marginalises).  Every factor of a window is invariant under a global translation and
     * a rotation about gravity; what the window knows about those four directions is carried by the marginal prior alone --
     * the memory of the anchor prior (GraphManager.cpp:27-35), which decays with every marginalisation (measured on a
     * 200-keyframe window: 2e-3, 1e-4, 1e-5, 8e-7 after 100, 500, 1 000, 2 000 updates) until it is below the float64 rounding
     * of the 1e9-scale entries beside it: H becomes indefinite, LM trials are rejected at random (2 000+ updates), then solves
     * fail (3 500 updates of 1 000-keyframe windows; profiles/r05_soak_*).  gauge_floor is the value below which the
     * window's own softest eigenvalues (those of H along its global translation / yaw) are not allowed to fall: at every
     * marginalisation the eigenvalues of the prior's 4 x 4 gauge information below gauge_floor * n / 3 (n keyframes in the
     * window: a unit gauge vector of the window has 3 / n of its weight on the three keyframes the prior touches) are lifted
     * back to that.  Default 3e-4, two orders above the rounding of H's largest entries (2e-6): for 1 000 keyframes an
     * information of 0.1 on WHERE the window is -- a 3 m sigma; it constrains nothing the factors can see, and a prior whose
     * gauge information is still above it (the first tens of updates) is not touched, bit for bit.  0 = off. */
    double gauge_floor;
    /* Incremental updates: the banded form of what ISAM2::update does with relinearizeThreshold / relinearizeSkip 1
     * (GraphManager.cpp:37-43,126-127).  != 0: vf_engine_isam_step relinearises only the keyframes whose pending increment
     * reaches the threshold, and linearises, assembles and eliminates again only from the first keyframe that moved or whose
     * factors are new: forward elimination is causal in time, so the Cholesky panels in front of it stand; the sweep restarts
     * from a checkpoint of its trailing window (one per 8 keyframe slots, 0.77 KB per slot) and the back substitution stops
     * once three consecutive increments come out as they were.  The cost of an update then follows the number of keyframes
     * it touches, not the length of the window.  Same arithmetic in the same order as the one-wave whole-window sweep: with
     * wildfire = 0 the increments equal that sweep's to the bit.  Windows that hold far factors, and refining engines, take the
     * full update.  2 = the same kernels started at the window's first keyframe every time (what the incremental update is
     * checked against, bit for bit: tests/test_gpu_incremental.py).  Default 0.
     * MEASURED (DESIGN.md "Incremental updates"): on the synthetic Carla stream one update moves the estimate of a keyframe
     * 1 500 keyframes back by 2-3 mm -- the IMU chain is stiff (2e-5 m), the odometry factors soft (0.3 m), so every update
     * re-estimates tilt and bias for the whole history -- and with the reference's threshold of 1e-4 nine keyframes in ten are
     * relinearised at every update: the suffix IS the window, here as in iSAM2.  The mode pays when the threshold is
     * loose or the graph has stiff odometry; it is off by default. */
    /* VF_ABI_TAIL: the fields from here to the end of the struct were added after the first sized release (tests/test_abi.py
     * builds a caller against the struct without them) */
    int incremental;
    double wildfire;         /* incremental updates: an increment that changes by at most this in every component counts as unchanged
                                (ISAM2Params::wildfireThreshold, 1e-3 in GTSAM); default 0 = bitwise */
    double min_model_fidelity; /* > 0: LM trials are accepted by GTSAM's rule instead of accept_rel
                                (LevenbergMarquardtOptimizer::tryLambda, the optimiser commented out at GraphManager.cpp:128-129): a trial
                                is accepted iff modelFidelity = (cost - new cost) / (cost - cost of the linearised problem at delta)
                                exceeds this (LevenbergMarquardtParams::minModelFidelity = 1e-3).  The prediction needs the gradient
                                in memory: such engines run the assembly kernel (K3), not the assembling sweep; not for time-sharded
                                windows.  Where the optimum is does not depend on the rule; trial counts and lambda histories do.
                                Default 0 = the library's own test. */
    int max_far_factors;     /* far between factors a window may hold at once (vf_engine_set_extra_between; the far ends of its linear
                                far factor count).  0 = VF_MAX_EXTRA, at most VF_MAX_FAR_LIMIT.  While VF_MAX_EXTRA or fewer are alive the
                                small dense systems of the low-rank correction and of the joint marginalisation live in LDS; beyond
                                (engines made for more) in device memory, 0.7 MB per window -- the same arithmetic in the same
                                order, so the same bits for the same factors, a few times slower per system.  Every far factor alive costs six
                                Woodbury columns per trial (single-window engines: six windows of the column engine, 23 MB each at
                                1 200 slots).  Arrays a caller passes for a window's far list (vf_engine_get_extra_between,
                                vf_engine_get_linear_far) hold this many entries. */
} vf_engine_opts;
/* Solver-form switches: how the library maps the solve onto the part, not what it computes.  vf_engine_default_tuning sets the
 * measured optimum and vf_engine_create uses exactly that; a binding of the reference never touches this struct.  They are
 * fields, not environment variables, so that the library's behaviour never depends on the caller's environment; tests and
 * tools/ set them (vf_engine_create_tuned) to reach one form on purpose.  Grows at its end only, like vf_engine_opts. */
typedef struct {
    uint32_t struct_size;
    int sweep_two_sided_max; /* whole-window sweeps: up to this many windows, two waves per window eliminate from both ends
                                (latency); above, one wave per window (throughput).  Default 256; 0 = always one wave. */
    int hybrid_threshold;    /* with vf_engine_set_convergence on a sweep engine of > 128 windows: once at most this many
                                windows still take trials, K4 runs as the partitioned form.  Default 256; < 0 = never. */
    int use_hip_graph;       /* != 0: vf_engine_iterate replays its launch sequence from a captured hipGraph, re-captured whenever
                                the trial count, the warm-start tail or a scalar baked into the kernel arguments changes
                                (vf_engine_graph_info counts captures and replays).  Bit-identical to plain launches
                                (tests/test_gpu_hip_graph.py); measured no faster, the stream never runs empty (bench.py
                                `single_window.with_hip_graph`); default 0 */
    int solve_split_min;     /* whole-window sweeps (one wave per window): from this many windows on, the forward sweep and the
                                back substitution run as two kernels (same arithmetic, same bits): the back substitution needs
                                9 KB of LDS instead of 38 and runs several waves per SIMD.  Pays once the batch is a multiple
                                of the 1024 SIMDs of the part.  Default 2048; 0 = never. */
    int solve_assemble_min;  /* whole-window sweeps: from this many windows on, the forward sweep forms the block rows of the
                                normal equations itself, from the Jacobians K1 / K2 leave, and the assembly kernel (K3) is not
                                launched: H is neither written nor read back (vf_engine_read_normal assembles it on demand).
                                Same sums in another order: agrees with the two-kernel form to rounding, not to the bit.  Not
                                used while a window holds far between factors, on sharded engines, or by the partitioned
                                form; in the hybrid solve (termination rule on) the sweep half uses it and the assembly
                                kernel runs for the partitioned half only.  Default 768 (below ~500 windows the assembly
                                kernel's launch is shorter than what the sweep pays, and the sweep pays it for rejected
                                trials too); 0 = never. */
    int solve_assemble_waves; /* 1: the assembling sweep is one wave per window; 2 (default): two waves per window sharing its
                                LDS, one eliminating, one assembling the rows -- the same bits, 6-10 % faster; launched in
                                chunks of 1024 windows, the workgroups the part holds at once */
    int hybrid_active_list;  /* hybrid solves (termination rule on, > 128 windows): the one-wave sweeps take their windows from a compacted
                                list of those still taking trials, so that the active ones are dispatched first (default 1; 0 = window i
                                is workgroup i as before; same bits either way) */
    int far_batch_columns;   /* single-window engines (the GraphManager's) holding far factors: the 6 Woodbury columns per far factor
                                are solved as ONE batch on a second, internal engine -- a copy of the window's H per column, 6 windows
                                per far factor alive (made for 1, 2, 4, 8, 16, 32 factors as they come: 0.12 KB per keyframe slot and window,
                                0.14 GB and 2 ms for the first loop closure of a 1 200-slot handle) -- instead of one band solve after
                                the other (default 1; 0 = sequential columns as on batch engines; same bits) */
    int far_big_forms;       /* engines made with max_far_factors > VF_MAX_EXTRA: the small dense systems of the far factors in device
                                memory even while VF_MAX_EXTRA or fewer are alive (default 0 = the LDS forms then; same bits: what
                                tests/test_gpu_far_capacity.py compares) */
} vf_engine_tuning;


/* Asynchronous staging (the GraphManager handle's engine runs this way; one-window engines that own their stream): with on != 0,
 * vf_engine_preintegrate (of keyframes beyond the window's end), vf_engine_set_between (up to 4 records),
 * vf_engine_marginalize, vf_engine_drop_oldest and vf_engine_set_range enqueue their work and return without waiting for the
 * device -- the marginalisation on a second stream, beside the staging of the keyframe that arrives (with far factors alive: on the
 * main stream, and the second one carries the Woodbury columns beside the window's own band solve) -- and what the device finds
 * wrong is reported by vf_engine_read_result instead of by the call: device_flags bit 0 = a preintegrated covariance was not
 * positive definite (VF_ERR_NOT_SPD of vf_engine_preintegrate), bit 1 = the pivot block of a marginalised keyframe was not,
 * bit 2 = the far ends' block (VF_ERR_INDETERMINATE of vf_engine_marginalize); the flags are cleared by the read.
 * vf_engine_read_result: the state of keyframe `slot` (estimate != 0: theta (+) delta of the reference-compat solve), the cost and the
 * LM counters of vf_engine_read_lm, in ONE synchronisation -- none at all when it follows a vf_engine_iterate whose adaptive trial
 * loop (one window under the termination rule) has read that very block to see the rule's flag up; any output may be null.  Works
 * on every engine. */
int vf_engine_set_async(vf_engine* e, int on);
/* asynchronous engines, after a solve: compute NOW, behind the solve, the marginal prior the next vf_engine_marginalize of this
 * window will need (its inputs -- the linearisation of the factors on the oldest keyframe at the solved states -- are final), so
 * that the call itself only puts it in place.  Used if nothing but appends happens in between; dropped otherwise.  The
 * GraphManager handle does this whenever its fixed-lag window is full.  A no-op on engines it does not apply to. */
int vf_engine_marginalize_ahead(vf_engine* e);
int vf_engine_read_result(vf_engine* e, int window, int slot, int estimate, double* state16, double* cost, int* accepted, int* rejected,
                          int* solve_failures, int* device_flags);
/* incremental engines (vf_engine_opts.incremental): updates made so far by vf_engine_isam_step, how many of them eliminated the
 * whole window (the first, and every one after an entry point the bookkeeping does not follow); for `window`, the keyframe slot
 * the last forward sweep started at and the slot the last back substitution stopped at (-1 on other engines) */
int vf_engine_incremental_info(vf_engine* e, int window, long* updates, long* whole_window_updates, int* first_eliminated,
                               int* last_substituted);
/* vf_engine_default_opts(o) fills the sizeof(vf_engine_opts) bytes of the header this translation unit was compiled against (the
 * macro below hands that size to the library); bindings that mirror the struct by hand (ctypes, cgo, JNI) call the function of the
 * same name, which fills the library's own size, or vf_engine_default_opts_sized with theirs. */
void vf_engine_default_opts(vf_engine_opts* o);
void vf_engine_default_opts_sized(vf_engine_opts* o, uint32_t struct_size);
void vf_engine_default_tuning(vf_engine_tuning* t);
void vf_engine_default_tuning_sized(vf_engine_tuning* t, uint32_t struct_size);
#ifndef VF_NO_SIZED_DEFAULTS
#define vf_engine_default_opts(o) vf_engine_default_opts_sized((o), (uint32_t)sizeof(vf_engine_opts))
#define vf_engine_default_tuning(t) vf_engine_default_tuning_sized((t), (uint32_t)sizeof(vf_engine_tuning))
#endif
int vf_engine_create(const vf_engine_opts* o, vf_engine** out);
int vf_engine_create_tuned(const vf_engine_opts* o, const vf_engine_tuning* t, vf_engine** out);   /* t = NULL: the defaults */
void vf_engine_destroy(vf_engine* e);

/* host -> device staging (each converts AoS records to the AoSoA device layout) */
int vf_engine_set_range(vf_engine* e, int window, int lo, int hi); /* active keyframes [lo,hi) */
int vf_engine_set_states(vf_engine* e, int window, int k0, int n, const double* state16);
int vf_engine_get_states(vf_engine* e, int window, int k0, int n, double* state16);
/* imu factor slot k holds the CombinedImuFactor X(k-1),V(k-1),X(k),V(k),B(k-1),B(k)
 * (IMUManager.cpp:68-73) */
int vf_engine_set_imu(vf_engine* e, int window, int k0, int n, const double* rec190);
/* BetweenFactor<Pose3>(X(a), X(b)) (GraphManager.cpp:86); stored in the slot of b; a < b,
 * b - a <= bandwidth; at most one between factor may end at a keyframe. */
int vf_engine_set_between(vf_engine* e, int window, int n, const int32_t* a, const int32_t* b,
                          const double* rec28);
int vf_engine_clear_between(vf_engine* e, int window, int k0, int n);
/* "Far" between factors: BetweenFactor<Pose3> on ANY pair of keyframes a < b of a window -- wider than the band
 * (b - a > bandwidth), or a second factor ending at a keyframe that already has one (a loop closure).  iSAM2 takes any
 * pair of keys (GraphManager.cpp:83-88); the banded device solver keeps such factors out of H and applies them as a
 * low-rank correction: g gets their J^T r, and every LM trial solves (H_band + lambda I + U U^T) delta = -g by Woodbury
 * -- the band solver once more per column of U (6 per factor) and one small dense system per window.  A FALLBACK for
 * the rare window with such factors, several times slower than a band-only window; at most VF_MAX_EXTRA per window
 * (vf_engine_opts.max_far_factors raises that to VF_MAX_FAR_LIMIT).
 * The call REPLACES the window's list (n = 0 clears it); records as for vf_engine_set_between.
 * A far factor outlives the keyframe it is anchored on, as in the reference's unbounded graph (GraphManager.cpp:83-88).  When
 * its older keyframe a is MARGINALISED (vf_engine_marginalize / vf_engine_slide(.., 1)) the factor is marginalised with it,
 * exactly at the current linearisation: it leaves this list and joins the window's LINEAR far factor -- six whitened rows
 * per far end over the three keyframes of the marginal prior and all the far ends (marginalising a keyframe that several far
 * factors touch couples their far ends: they are one factor from then on) -- which every later marginalisation re-expresses
 * the same way, and whose far ends are folded into the marginal prior as they come within its reach (the leaving keyframe
 * + 3): the information of a loop closure outlives BOTH its ends, and a fixed-lag window follows the whole-history optimum
 * across them to 1e-7 m (tests/test_gpu_far_factors.py).  (Far ends count against VF_MAX_EXTRA; vf_engine_get_linear_far
 * lists them.)  A slide that does not marginalise re-anchors the
 * factor on keyframe a + 1 instead -- Z' = D^-1 Z with D the current estimate of T_a^-1 T_a+1, taken as exact -- and drops
 * the linear ones, as it drops the band factors of the keyframe that leaves.  Engines holding far factors start every solve
 * cold.  Not for time-sharded engines. */
int vf_engine_set_extra_between(vf_engine* e, int window, int n, const int32_t* a, const int32_t* b, const double* rec28);
/* the window's (nonlinear) far factors as they stand, and how many far factors were transported (made linear by a
 * marginalisation, or re-anchored by a slide without one) / dropped by a slide that does not marginalise / absorbed into the
 * marginal prior over the life of the engine; any output pointer may be NULL */
int vf_engine_get_extra_between(vf_engine* e, int window, int* n, int32_t* a, int32_t* b, double* rec28, long* transported, long* ended,
                                long* absorbed);
/* the window's linear far factors: their number and the window-local keyframe each one ends at (far_end: max_far_factors ints -- VF_MAX_EXTRA by default -- or NULL) */
int vf_engine_get_linear_far(vf_engine* e, int window, int* n, int32_t* far_end);
/* the three priors of GraphManager.cpp:27-35 as one diagonal 15-row factor on keyframe k */
int vf_engine_set_prior(vf_engine* e, int window, int k, const double* rec31);

/* K0: IMU preintegration on the device.  Factor i (keyframe k0+i) integrates
 * steps[step_off[i] .. step_off[i+1]) (7 doubles each: dt, acc xyz, gyro xyz) with bias estimate
 * bias_hat6[i]; the result (mean, bias Jacobians, R = chol_upper(preintMeasCov^-1)) lands in the
 * factor's device record.  Replaces PreintegratedCombinedMeasurements::resetIntegrationAndSetBias
 * + integrateMeasurement as driven by IMUManager::getFactor (IMUManager.cpp:42-64). */
typedef struct {
    double acc_cov, gyro_cov, integration_cov;          /* ImuManagerRos.cpp:22-24,30-32 */
    double bias_acc_cov, bias_omega_cov, bias_acc_omega_int; /* ImuManagerRos.cpp:20-21,25,28-29,33 */
} vf_imu_params;
int vf_engine_preintegrate(vf_engine* e, int window, int k0, int n, const int32_t* step_off,
                           const double* steps7, const double* bias_hat6, const vf_imu_params* p);
int vf_engine_get_imu(vf_engine* e, int window, int k0, int n, double* rec190);
/* The INGEST half of a fixed-lag update, for every window at once -- what GraphManager::reserveNode (GraphManager.cpp:51-69)
 * and addBetweenFactor (:83-88) do for one vehicle per keyframe: window w's next IMU factor (hi[w]-1 -> hi[w]) is
 * preintegrated on the device from its raw samples steps7[step_off[w] .. step_off[w+1]) WITH THE WINDOW'S CURRENT BIAS
 * ESTIMATE (the bias of its last keyframe, read on the device: `getFactor(_lastPoseTime, time, _currentKey, getBias())`,
 * GraphManager.cpp:59), and the between factor that ends at the same keyframe is staged (btw_a[w] = source keyframe
 * slot, -1 = none; btw_rec28[w] its record).  One packed host->device copy from a pinned buffer, one kernel launch, no host
 * synchronisation; vf_engine_slide then appends the keyframe (its initial value predicted from this factor) and
 * vf_engine_iterate solves, the engine staying warm.  Errors found on the device (no free slot, covariance not SPD) are
 * reported by vf_engine_ingest_status, which also returns the last call's copy / kernel times from HIP events. */
int vf_engine_ingest_tail(vf_engine* e, const int32_t* step_off, const double* steps7, const vf_imu_params* p,
                          const int32_t* btw_a, const double* btw_rec28);
int vf_engine_ingest_status(vf_engine* e, float* h2d_ms, float* k0_ms);

/* ---- hot-path stages (asynchronous on the engine's HIP stream) ---- */
/* K1+K2+priors: residual + whitened Jacobian of every factor, at the current (which=0) or
 * trial (which=1) states.  Replaces the linearisation inside ISAM2::update
 * (GraphManager.cpp:126). */
int vf_engine_linearize(vf_engine* e, int which);
/* K3: block-banded J^T J, J^T r of the current linearisation */
int vf_engine_assemble(vf_engine* e);
/* K4: (H + lambda I) delta = -g by block-banded Cholesky, one wavefront per window */
int vf_engine_solve(vf_engine* e);
/* K5: trial = current (+) delta ; then linearize(which=1) + accept/reject */
int vf_engine_retract(vf_engine* e);
int vf_engine_decide(vf_engine* e, int init);
/* `iterations` LM trials (linearize once, then {assemble, solve, retract, linearize(trial),
 * decide} per trial).  Replaces ISAM2::update + calculateEstimate (GraphManager.cpp:126-127).
 * Warm start: when nothing but vf_engine_slide has touched the engine since the previous
 * vf_engine_iterate, the opening linearisation covers only the appended keyframes' factors and
 * the priors, and the first assembly only the ends of windows whose last trial was rejected --
 * every other record, H row and g entry is still the one of the current states.  Results are
 * bit-identical to a cold start (which any other mutating call brings back). */
int vf_engine_iterate(vf_engine* e, int iterations);
/* ---- time-sharded windows (one window spread over the GPUs of a node; no reference code: the reference
 * is a single process.  SURVEY.md 8e / BASELINE.json configs[4]) ----
 * Every rank holds the whole window (states and factors replicated) in an engine created with the same
 * explicit chunk count (vf_engine_opts.chunks = P, a multiple of the world size).  Rank r owns the chunks
 * [r P / world, (r+1) P / world) and the keyframes they cover: vf_engine_linearize then writes the Jacobians
 * of the owned keyframes' factors only (plus a 5-slot halo) but the residual of EVERY factor, vf_engine_assemble
 * works on the owned rows, and one LM trial is
 *     assemble, solve_local, <all-gather sep>, solve_global, <all-reduce delta>, retract, linearize(trial), decide
 * i.e. TWO collectives, both the caller's (RCCL over xGMI via torch.distributed in
 * vil_sensor_fusion_amd/distributed.py; the library itself has no communication dependency):
 *   sep    the packed separator system, one slot of 2248 doubles per (chunk, window) = [27x28 | 27x28 | 27x27] + pad,
 *          chunk-major, so a rank's chunks are one contiguous slice: an in-place all-gather;
 *   delta  15 doubles per keyframe slot (non-owned entries zero) followed by one solve-failure flag per window: a sum.
 * The cost of a trial needs no exchange: every rank has every residual and takes the same accept / reject decision. */
typedef struct {
    int rank, world, windows, chunks;
    void* sep;              /* device, [chunks][windows][2248]: rank r writes chunks [r P/world, (r+1) P/world) */
    long sep_per_chunk;     /* doubles per chunk in sep (= windows * 2248) */
    void* delta;            /* device: increments of all keyframe slots (non-owned entries are zero after solve_global), */
    long delta_count;       /*         then one failure flag per window: capacity * windows * 15 + windows doubles */
    void* refine_delta;     /* device, shaped like delta: what vf_engine_solve_global leaves while a refinement is open (NULL when
                               vf_engine_opts.refine_iterations == 0) */
} vf_shard_info;
/* Geometry of the partitioned solve, host only (no device needed): an n-keyframe window is cut into `count`
 * chunks (<= chunks; fit != 0: also <= sqrt(n)); chunk c = `interior` keyframes from window-local
 * keyframe `first`, followed by 3 separator keyframes when has_separator.  vf_shard_range: the chunks
 * [chunk_lo, chunk_hi) and window-local keyframes [kf_lo, kf_hi) rank `rank` of `world` owns. */
int vf_chunk_geometry(int n, int chunks, int fit, int c, int* count, int* first, int* interior, int* has_separator);
int vf_shard_range(int n, int chunks, int fit, int rank, int world, int* chunk_lo, int* chunk_hi, int* kf_lo, int* kf_hi);
/* run every later stage on the caller's HIP stream (hipStream_t), e.g. the one its collectives use */
int vf_engine_set_stream(vf_engine* e, void* hip_stream);
int vf_engine_set_shard(vf_engine* e, int rank, int world);
int vf_engine_shard_info(vf_engine* e, vf_shard_info* out);
int vf_engine_solve_local(vf_engine* e);    /* chunk sweeps + spikes of the owned chunks */
int vf_engine_solve_global(vf_engine* e);   /* separator chain, back substitution of the owned chunks, zero the rest of delta */
int vf_engine_reset_lambda(vf_engine* e);   /* lambda := lambda0, as vf_engine_iterate does before its first trial */
/* The refined solve (vf_engine_opts.refine_iterations) on a time-sharded engine, staged like the solve: after the trial's two
 * collectives,  vf_engine_refine_begin ; N x { vf_engine_solve_local, <all-gather sep>, vf_engine_solve_global, <all-reduce
 * refine_delta>, vf_engine_refine_step } ; vf_engine_refine_end -- vf_engine_solve_local / _global work on the correction's
 * right-hand side while a refinement is open; N = vf_engine_refine_count (0: this engine does not refine now; the same on
 * every rank).  Unsharded engines do all of it inside vf_engine_solve. */
int vf_engine_refine_count(vf_engine* e, int* iterations);
int vf_engine_refine_begin(vf_engine* e);
int vf_engine_refine_step(vf_engine* e);
int vf_engine_refine_end(vf_engine* e);
/* corrections the last refined solve of `window` applied, and res . M^-1 res at its end relative to its first value */
int vf_engine_read_refine(vf_engine* e, int window, int* corrections, double* reduction);

/* The same from C (RCCL): the library issues the collectives itself, on the engine's stream, through the communicator the
 * caller hands in -- an ncclComm_t created by the caller (ncclCommInitRank: one rank per GPU), passed as void* so that this
 * header needs no RCCL header.  librccl is looked up with dlopen when one of these is first called: the library has no
 * link-time communication dependency.  vf_shard_iterate = vf_engine_iterate for a time-sharded engine (every rank calls it
 * with the same arguments; 2 x (1 + refinement corrections) collectives per LM trial); vf_shard_gn_step = vf_engine_isam_step.
 * vf_shard_exchange_plan (host only): what a rank sends -- its slice [sep_offset, sep_offset + sep_count) of the sep buffer
 * of sep_total doubles, all-gathered in place, and delta_count doubles all-reduced -- for callers that bring another
 * transport (vil_sensor_fusion_amd/distributed.py exchanges the same ranges through torch.distributed). */
int vf_shard_iterate(vf_engine* e, void* nccl_comm, int iterations);
int vf_shard_gn_step(vf_engine* e, void* nccl_comm, double relin_threshold);
int vf_shard_exchange_plan(int windows, int capacity, int chunks, int rank, int world, long* sep_offset, long* sep_count,
                           long* sep_total, long* delta_count);

/* Reference-compat solve: what the reference computes per GraphManager::solve (GraphManager.cpp:38-43,126-127) -- ONE
 * iSAM2-like update: keyframes whose pending increment reaches relin_threshold in any component (ISAM2Params::
 * relinearizeThreshold, 1e-4 in the reference) move their linearisation point there, every factor is linearised at the
 * linearisation points, ONE undamped Gauss-Newton system is solved for the increments of all keyframes, and the
 * estimate is theta (+) delta.  vf_engine_get_states then returns the linearisation points, vf_engine_get_estimate the
 * estimate; vf_engine_predict_from_estimate starts the IMU prediction of new keyframes from the estimate of keyframe
 * k0 - 1 (GraphManager.cpp:152-153).  Exact where iSAM2 is approximate (its partial back-substitution stops below the
 * wildfire threshold).  Not for sharded engines; fixed-lag marginalisation is not part of this mode. */
int vf_engine_isam_step(vf_engine* e, double relin_threshold);
/* the opening of such an update on its own -- relinearise, lambda := 0, linearise at theta -- for callers that stage the
 * solve themselves (time-sharded engines: vf_engine_assemble, vf_engine_solve_local, ..., vf_engine_retract follow) */
int vf_engine_gn_begin(vf_engine* e, double relin_threshold);
int vf_engine_predict_from_estimate(vf_engine* e, int window, int k0, int n);
int vf_engine_get_estimate(vf_engine* e, int window, int k0, int n, double* state16);

/* Optional LM termination (off by default: vf_engine_iterate runs exactly `iterations` trials).  With a
 * tolerance > 0, a window whose trial changes the cost by <= abs_tol, or by <= rel_tol * cost (an accepted
 * step that no longer pays, or a rejected one inside the rounding floor), is converged and takes no part in
 * the remaining trials of that solve: the rule of
 * gtsam::LevenbergMarquardtOptimizer (checkConvergence; LevenbergMarquardtParams defaults 1e-5 / 1e-5), the
 * optimiser commented out at GraphManager.cpp:128-129.  (0, 0) switches it off again. */
int vf_engine_set_convergence(vf_engine* e, double rel_tol, double abs_tol);

/* Fixed-lag marginalisation of every window's oldest keyframe (no reference code: the
 * reference's iSAM2 graph is unbounded): Schur complement of all factors touching it, at the
 * current linearisation, into a dense Gaussian prior on [next: 15 dof][next+1: pose][next+2:
 * pose].  Needs >= 4 keyframes per window and a preceding vf_engine_iterate / linearize. */
int vf_engine_marginalize(vf_engine* e);
/* lo += 1 on every window (call after vf_engine_marginalize) */
int vf_engine_drop_oldest(vf_engine* e);
/* slide every window by one keyframe: hi += 1 (new keyframe's state predicted from its IMU
 * factor, GraphManager.cpp:152-160), lo += 1.  marginalize != 0: the dropped keyframe is
 * marginalised into the dense prior above; marginalize == 0: the priors of GraphManager.cpp:27-35
 * are re-anchored on the new oldest keyframe at its current estimate (conditioning, overconfident). */
int vf_engine_slide(vf_engine* e, const double* prior_sigma15, int marginalize);
int vf_engine_read_marginal(vf_engine* e, int window, int* on, double* xbar48, double* L729, double* eta27);
int vf_engine_predict(vf_engine* e, int window, int k0, int n); /* states k0..k0+n-1 from k-1 */
/* Reclaim keyframe slots: move the live keyframes [shift, capacity) of every window down to slot 0
 * (states, factor records, linearisations, between sources, ranges, prior keys).  shift must be
 * a multiple of 64 and <= every window's lo.  Lets a fixed-lag smoother run indefinitely. */
int vf_engine_compact(vf_engine* e, int shift);
/* Grow every window to `new_capacity` keyframe slots (> the current capacity; rounded up to a multiple of 64): states
 * (both buffers), pending increments, factor records, between sources, priors / marginal priors, ranges and LM counters
 * are carried over on the device; the linearisation of the current states is recomputed, normal equations are not kept
 * (the next solve starts cold).  This is
 * what lets a GraphManager with lag = 0 keep the whole history the way the reference's unbounded iSAM2 graph does
 * (GraphManager.cpp:17-43), for as long as the 24 KB per keyframe slot fit in HBM.  Not for sharded engines.
 * Memory: the old and the new buffers are alive together until the copies are done -- a doubling peaks at 3x the old
 * footprint.  Cost: 36 pitched device-to-device copies (32 state planes + 4 arrays) whatever the number of windows. */
int vf_engine_grow(vf_engine* e, int new_capacity);
int vf_engine_sync(vf_engine* e);
/* vf_engine_opts.use_hip_graph: is replay active (0 once a capture failed or a caller's stream was handed in), how often the
 * launch sequence was captured, how many vf_engine_iterate calls were served by hipGraphLaunch */
int vf_engine_graph_info(vf_engine* e, int* enabled, int* captures, long* replays);
/* which form of K4 the next vf_engine_solve launches (it follows from the options, the batch and what the windows hold):
 * 0 one wave per window, one kernel; 1 the same as forward sweep + back substitution (solve_split_min); 2 the assembling
 * forward sweep + back substitution, no K3 (solve_assemble_min); 3 two waves per window from both ends; 4 partitioned;
 * 5 hybrid of the one-wave and the partitioned form (termination rule on) */
int vf_engine_solve_form(vf_engine* e, int* form);

/* ---- read-back (synchronises) ---- */
int vf_engine_read_imu_lin(vf_engine* e, int window, int which, int k0, int n, double* r15, double* J450);
int vf_engine_read_between_lin(vf_engine* e, int window, int which, int k0, int n, double* r6,
                               double* Ja36, double* Jb36);
int vf_engine_read_normal(vf_engine* e, int window, int k0, int n, double* Hband, double* g15);
int vf_engine_read_delta(vf_engine* e, int window, int k0, int n, double* delta15);
/* Cholesky panels of the last vf_engine_solve: per keyframe 43 rows x 16 doubles (15 used) =
 * rows of L in the keyframe's 15 columns for [k+1:15][k+2:pose 6][k+3:pose 6] (27 rows), the
 * forward-substituted rhs row (1), and L_kk^-T (15 rows, upper triangular). */
int vf_engine_read_panels(vf_engine* e, int window, int k0, int n, double* panels688);
int vf_engine_read_lm(vf_engine* e, int window, double* cost, double* lambda, int* accepted,
                      int* rejected, int* solve_failures);
/* non-monotone LM: trials kept provisionally so far (not counted in accepted / rejected), and whether an excursion is open */
int vf_engine_read_excursions(vf_engine* e, int window, int* provisional_trials, int* open_now);
/* non-monotone LM: undo an excursion that is still open (restore the point it started from).  vf_engine_iterate does this
 * after its last trial; callers that stage their trials (time-sharded windows) call it after theirs. */
int vf_engine_close_excursions(vf_engine* e);

/* ---- measurement: time `reps` launches of one stage with HIP events on the engine stream ---- */
#define VF_STAGE_LINEARIZE_IMU 1
#define VF_STAGE_LINEARIZE_BTW 2
#define VF_STAGE_ASSEMBLE 3
#define VF_STAGE_SOLVE 4
#define VF_STAGE_RETRACT 5
#define VF_STAGE_DECIDE 6
#define VF_STAGE_ASSEMBLE_IDLE 7   /* K3 when every window's last trial was rejected: nothing to assemble, the cost of its launch */
int vf_engine_time_stage(vf_engine* e, int stage, int reps, float* avg_ms);
/* HIP-event time of a whole vf_engine_iterate(iterations) */
int vf_engine_time_iterate(vf_engine* e, int iterations, float* ms);
int vf_engine_counts(vf_engine* e, int64_t* n_imu, int64_t* n_between, int64_t* n_keyframes);

/* ===================================================================== degeneracy metrics (K6)
 * Batched form of the reference's per-message metric calls:
 *   y[0] = 0 ; y[i] = metric(mat_now = mats[i], mat_prev = mats[i-1], pose_now, pose_prev)
 * (apply_degen_function, vil_fusion/python/make_prettier_graphs.py:547-576).
 * mats: (count,6,6) row-major; pose: (count,6) [x y z roll pitch yaw] or NULL; dtype 0 = f64,
 * 1 = f32 (mats/pose/out all of that type); subset 0 = all 6x6, 1 = trans [0:3,0:3],
 * 2 = rot [3:6,3:6].  metric ids follow degen_funcs
 * (vil_fusion/python/degeneracy_detection_functions.py:283-303) then condition_number,
 * differential_entropy.  reps/kernel_ms: optional HIP-event timing of the kernel alone. */
#define VF_METRIC_D_OPT 0
#define VF_METRIC_D_OPT_RATIO 1
#define VF_METRIC_A_OPT 2
#define VF_METRIC_A_OPT_RATIO 3
#define VF_METRIC_E_OPT 4
#define VF_METRIC_E_OPT_RATIO 5
#define VF_METRIC_MAX_EIGEN 6
#define VF_METRIC_MAX_EIGEN_RATIO 7
#define VF_METRIC_JENSEN_BREGMAN 8
#define VF_METRIC_CORRELATION_MATRIX_DISTANCE 9
#define VF_METRIC_KULLBACK_LEIBLER 10
#define VF_METRIC_NORM_FROBENIUS 11
#define VF_METRIC_NORM_FROBENIUS_RATIO 12
#define VF_METRIC_NORM_NUCLEAR 13
#define VF_METRIC_NORM_NUCLEAR_RATIO 14
#define VF_METRIC_NORM_1 15
#define VF_METRIC_NORM_1_RATIO 16
#define VF_METRIC_NORM_2 17
#define VF_METRIC_NORM_2_RATIO 18
#define VF_METRIC_CONDITION_NUMBER 19
#define VF_METRIC_DIFFERENTIAL_ENTROPY 20
int vf_degeneracy_batch(const void* mats, const void* pose, int count, int dtype, int subset, int metric,
                        void* out, int reps, float* kernel_ms);
/* extra: the three metrics that read the ends of the spectrum -- e_opt, max_eigen, condition_number
 * (degeneracy_detection_functions.py:74-82, 98-106, 239-243) -- of ONE eigen-solve per matrix, one launch, three outputs (each `count`
 * values, [0] = 0): bit for bit what three vf_degeneracy_batch calls return, at a third of the work.  condition_number is NaN for a
 * matrix that is not symmetric to rounding (its singular values are then not the moduli of its eigenvalues: ask
 * vf_degeneracy_batch, which falls back to an SVD). */
int vf_degeneracy_spectrum_batch(const void* mats, int count, int dtype, int subset, void* e_opt, void* max_eigen,
                                 void* condition_number, int reps, float* kernel_ms);
/* The shipped gate (gtsam_fusion/src/degerate_odometry_filter.cpp:29-47): float32 log det of the
 * rotation (3,3) and translation (0,0) 3x3 blocks of the 36-float LOAM Hessian; keep[i] = 0 when
 * either is below its threshold (fusion_params.yaml:35-36: 11.5 / 28.9). */
int vf_dopt_filter_f32(const float* hessians36, int count, float rot_thr, float trans_thr, float* rot_dopt,
                       float* trans_dopt, unsigned char* keep);

/* ===================================================================== GraphManager surface
 * Drop-in for VILFusion::GraphManager + VILFusion::IMUManager.  ROS-free, usable in simulated
 * time, every call thread-safe (GraphManager.h:4-6): the same two-lock discipline as the
 * reference (_graphMutex for ingestion, _stateMutex for the estimate; GraphManager.h:103-104),
 * vf_solve releases the ingestion lock before optimising and runs callbacks on the solving
 * thread inside the state lock (GraphManager.cpp:104-138). */
typedef struct vf_graph vf_graph;

typedef struct {
    uint32_t struct_size; /* sizeof(vf_graph_opts) in the caller's header (vf_graph_default_opts sets it); same rule as vf_engine_opts */
    int capacity;    /* keyframe slots on the device (keys 0..capacity-1), rounded up to a multiple of 64 by vf_create (the
                        engine allocates whole tiles); with lag = 0 the INITIAL number: vf_solve doubles it whenever the
                        history outgrows it (vf_engine_grow), unless fixed_capacity != 0 */
    int lag;         /* fixed-lag window length in keyframes; 0 = smooth the whole history */
    int iterations;  /* LM trials per vf_solve, at most (see rel_tol / abs_tol) */
    int device;
    double prior_sigma[15]; /* X0/V0/B0 prior sigmas; default GraphManager.cpp:27-31 */
    /* LM termination inside a vf_solve (vf_engine_set_convergence): the solve stops taking trials once one changes the
     * cost by <= abs_tol or <= rel_tol * cost.  Defaults 1e-5 / 1e-5 = gtsam::LevenbergMarquardtParams (the optimiser at
     * GraphManager.cpp:128-129); 0 / 0 = always `iterations` trials. */
    double rel_tol, abs_tol;
    int cold_start;  /* != 0: the handle's engine never warm-starts a solve (vf_engine_opts.cold_start); default 0 */
    int fixed_capacity; /* != 0 with lag = 0: vf_reserve_node fails with VF_ERR_CAPACITY once `capacity` keyframes exist
                           (the behaviour before the history could grow); default 0 */
    /* reference_compat != 0: vf_solve does what the reference's solve() does -- one iSAM2-like update
     * (vf_engine_isam_step with relin_threshold, default 1e-4 = GraphManager.cpp:40) instead of LM to convergence; needs
     * lag == 0 (the reference's graph is unbounded).  Default 0. */
    int reference_compat;
    double relin_threshold;
    /* incremental != 0 (with reference_compat): the update is done incrementally (vf_engine_opts.incremental -- read its
     * MEASURED note -- and .wildfire): a vf_solve costs what the keyframes it touches cost.  Default 0. */
    /* VF_ABI_TAIL: as in vf_engine_opts */
    int incremental;
    double wildfire;
    double min_model_fidelity; /* vf_engine_opts.min_model_fidelity (GTSAM's LM accept rule; 1e-3 there); default 0 = the library's own */
    int synchronous_staging;   /* != 0: the handle's engine stages synchronously, as before round 6 (every staging call waits for the
                                  device and reports its own failures; no preintegration at vf_reserve_node, no marginal prior computed
                                  ahead): same bits as the default, slower; what tests compare the asynchronous path with.  Default 0 */
    int max_far_factors;       /* vf_engine_opts.max_far_factors: loop closures (between factors the band cannot hold) alive at once;
                                  vf_add_between returns VF_ERR_CAPACITY beyond.  0 = the default, VF_MAX_FAR_LIMIT (a handle has no
                                  caller-sized arrays to keep small; while VF_MAX_EXTRA or fewer are alive it solves with the LDS forms) */
} vf_graph_opts;

/* (time, pose q_wxyz, position, velocity, bias[acc,gyro]) -- GraphManager::OptimizationCallback
 * (GraphManager.h:38) */
typedef void (*vf_callback)(void* user, double time, const double q[4], const double t[3],
                            const double v[3], const double bias[6]);

void vf_graph_default_opts(vf_graph_opts* o);
void vf_graph_default_opts_sized(vf_graph_opts* o, uint32_t struct_size);
#ifndef VF_NO_SIZED_DEFAULTS
#define vf_graph_default_opts(o) vf_graph_default_opts_sized((o), (uint32_t)sizeof(vf_graph_opts))
#endif
/* GraphManager::GraphManager(imuManager) + IMUManager::IMUManager(params) + getImuParams
 * (GraphManager.cpp:15-44, IMUManager.cpp:13-17, ImuManagerRos.cpp:14-36) */
int vf_create(const vf_imu_params* imu, const vf_graph_opts* opts, vf_graph** out);
void vf_destroy(vf_graph* g);
/* extra (no reference twin): the reference anchors X(0) = identity, V(0) = 0, B(0) = 0 with the priors of
 * GraphManager.cpp:20-35, which suits a vehicle that starts level and at rest; to replay a log that starts in motion, put the
 * anchor (state16: q t v bias) and the means of the three priors somewhere else.  Only before the first vf_reserve_node. */
int vf_set_initial_state(vf_graph* g, const double state16[16]);
/* IMUManager::addIMUMeasurement (IMUManager.cpp:19-25) */
int vf_add_imu(vf_graph* g, double time, const double acc[3], const double gyro[3]);
/* GraphManager::reserveNode (GraphManager.cpp:51-69): 1-based key, key 0 is the prior node */
int vf_reserve_node(vf_graph* g, double time, uint64_t* key_out);
/* GraphManager::addBetweenFactor (GraphManager.cpp:83-88) with noiseModel::Gaussian::Covariance
 * (SensorManagerRos.cpp:99); cov is 6x6 row-major in Pose3 tangent order [rot, trans].  Any pair of reserved keys prev < cur: a
 * factor the band cannot hold (wider than VF_MAX_BANDWIDTH keyframes, or a second one ending at cur: a loop closure) is a far
 * factor (vf_engine_set_extra_between), of which vf_graph_opts.max_far_factors may be alive at once -- VF_ERR_CAPACITY beyond. */
int vf_add_between(vf_graph* g, uint64_t prev_key, uint64_t cur_key, const double q_wxyz[4],
                   const double t[3], const double cov36[36]);
/* GraphManager::addFactor(const CombinedImuFactor&) (GraphManager.cpp:90-94): queue a READY-MADE preintegrated factor
 * X(key-1),V(key-1),X(key),V(key),B(key-1),B(key) as its 190-double record (layout at the top of this file; e.g. what
 * vf_get_imu_factor returns) instead of cutting one from the IMU buffer.  key must be the next key (current + 1); the
 * node time advances by the record's deltaTij. */
int vf_add_imu_factor(vf_graph* g, uint64_t key, const double* rec190);
/* GraphManager::solve (GraphManager.cpp:101-141).  With a fixed lag, a between factor (band or far) added after its older
 * key has left the window is late odometry: the solve drops it, returns VF_ERR_BAD_KEY once, gives everything else it had
 * taken back to the queues, and the next vf_solve runs on the rest. */
int vf_solve(vf_graph* g);
/* GraphManager::getState / getBias / getMostRecentPoseTime (GraphManager.cpp:164-178, 71-75) */
int vf_get_state(vf_graph* g, double q[4], double t[3], double v[3], double bias[6]);
int vf_get_bias(vf_graph* g, double bias[6]);
int vf_most_recent_pose_time(vf_graph* g, double* time, uint64_t* key);
/* GraphManager::getMostRecentEstimate (GraphManager.cpp:77-81).  The reference never assigns the member it returns
 * (_mostRecentEstimate, GraphManager.h:109), so this is the default NavState -- identity pose, zero velocity -- always;
 * use vf_get_state for the estimate. */
int vf_get_most_recent_estimate(vf_graph* g, double q[4], double t[3], double v[3]);
/* GraphManager::addOptimizationCallback (GraphManager.cpp:96-99) */
int vf_set_callback(vf_graph* g, vf_callback cb, void* user);
/* GraphManager::graph()->size(): factors staged since the last solve (3 priors at start +
 * between factors; IMU factors wait in the queue, GraphManager.cpp:66,150) and the queue length */
int vf_graph_staged(vf_graph* g, int* staged_factors, int* queued_imu_factors);
/* GraphManager::graph() (GraphManager.h:42, GraphManager.cpp:46-49), by index 0 .. staged_factors - 1: what the reference's tests
 * read from the staged NonlinearFactorGraph (test/UnitTests.cpp:200,222-233: its size, a factor's keys, measured()).
 * kind 0 / 1 / 2: PriorFactor<Pose3> on X(0) (q, t = its mean, cov36 = diag sigma^2 in [rot, trans] order), PriorFactor<Vector3>
 * on V(0) (t = mean, top-left 3 x 3 of cov36), PriorFactor<ConstantBias> on B(0) (t = accelerometer part, q[1..3] = gyro part,
 * q[0] = 0; cov36 6 x 6) -- all three on key1 = key2 = 0, present until the first solve takes them.  kind 3: BetweenFactor<Pose3>
 * X(key1) -> X(key2), measured() = (q, t) with q normalised as gtsam::Rot3(w, x, y, z) does, cov36 as handed in.  Any output
 * pointer may be null.  VF_ERR_BAD_KEY beyond the end. */
int vf_graph_get_staged(vf_graph* g, int index, int* kind, uint64_t* key1, uint64_t* key2, double q[4], double t[3],
                        double cov36[36]);
/* diagnostics (extra): cost after the last solve and the LM trials accepted / rejected / failed (normal equations not
 * positive definite) over the life of the handle */
int vf_graph_lm_stats(vf_graph* g, double* cost, int* accepted, int* rejected, int* solve_failures);
/* diagnostics (extra): keyframes in the window of the last solve; conjugate-gradient corrections every solve of the handle's
 * engine is now followed by (vf_engine_opts.refine_iterations: 0 until the window outgrows refine_min_keyframes -- a
 * whole-history handle, lag = 0, gets there by itself -- 12 from then on); LM trials kept provisionally so far (lm_excursion) */
int vf_graph_solver_info(vf_graph* g, int* window_keyframes, int* refine_corrections, int* provisional_trials);
/* diagnostics (extra; handles made with incremental != 0): updates so far, how many of them re-eliminated the whole history, and
 * the keys the last update's forward sweep started at / its back substitution stopped at */
int vf_graph_incremental_info(vf_graph* g, long* updates, long* whole_window_updates, uint64_t* first_eliminated_key,
                              uint64_t* last_substituted_key);
/* smoothed states of keys [key0, key0+n) after the last solve (extra; iSAM2 calculateEstimate) */
int vf_get_trajectory(vf_graph* g, uint64_t key0, int n, double* state16);
/* preintegrated record of the factor ending at `key` (extra; firstFactor->preintegratedMeasurements()) */
int vf_get_imu_factor(vf_graph* g, uint64_t key, double* rec190);

#ifdef __cplusplus
}
#endif
#endif
