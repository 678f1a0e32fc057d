// vf_kernels.hpp -- device view of an engine and the kernel launch interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vf {

constexpr int TILE = 64;        // AoSoA tile = one wavefront of factors
constexpr int IMU_IN = 190;     // dt, delta(9), bhat(6), H(54), R packed upper (120)
constexpr int IMU_OUT = 465;    // r(15), J(15x30 row-major): the documented (host-side / LDS) form of an IMU linearisation
// In HBM the whitened residual and the Jacobian of an IMU factor live apart:
//   imu_r  [2][G/64][15][64]   residuals, AoSoA like the inputs (the cost kernel reads nothing else)
//   imu_j  [2][G/8][JT_STRIDE] "J stream": per tile of 8 consecutive keyframe slots, the 291 entries of J that are not
//          structurally zero, in the order K1 produces them, i-side columns (147, padded to 74 pairs) then j-side columns
//          (144 = 72 pairs), stored as [pair][8 slots][2 doubles].  A K1 store instruction is 16 bytes per lane = 8 full
//          128-B lines (146 of them per factor instead of 291 8-byte ones); a K3 workgroup copies ONE contiguous 18.7 KB
//          block into LDS as it is, plus the i-side pairs of its halo factor (slot 0 of the next tile).
constexpr int IMU_R = 15;
#ifndef VF_JT_LOG
#define VF_JT_LOG 3
#endif
constexpr int JT_LOG = VF_JT_LOG, JT = 1 << JT_LOG; // keyframe slots per J tile (= K3's tile): 8
constexpr int JS_NI = 147, JS_NJ = 144;             // non-zero entries of the i-side / j-side columns
constexpr int JS_PI = (JS_NI + 1) / 2, JS_PJ = (JS_NJ + 1) / 2, JS_PAIRS = JS_PI + JS_PJ;   // 74 + 72 pairs
constexpr int JT_STRIDE = JS_PAIRS * JT * 2;        // 2336 doubles = 146 lines per tile
constexpr int BTW_IN = 28;      // q(4), t(3), R packed upper (21)
constexpr int BTW_OUT = 78;     // r(6), Ja(36), Jb(36)
constexpr int PLACE_CELLS = 2048;  // (xcc 3 bits, se / sh / cu 8 bits of HW_ID)
constexpr int MAX_EXTRA = 8;        // far between factors per window the LDS forms of k_marginalize / k_extra_combine hold (= VF_MAX_EXTRA)
constexpr int MAX_EXTRA_BIG = 32;   // ... and their global-scratch forms (= VF_MAX_FAR_LIMIT; vf_engine_opts.max_far_factors > VF_MAX_EXTRA)
// per-window scratch of those forms (View::far_scratch): W and E of the joint marginalisation, 6 X rows x (42 + 6 X + 1) columns
// each (the Woodbury system, 6 X x (6 X + 1), fits in one of them)
constexpr size_t FAR_SCRATCH = 2 * (size_t)(6 * MAX_EXTRA_BIG) * (42 + 6 * MAX_EXTRA_BIG + 1);
constexpr int PRIOR_IN = 31;    // mean state (16), sigma (15)
constexpr int PRIOR_OUT = 240;  // r(15), J(15x15)
// Block row of H of one keyframe k, as the solver reads it (512 doubles, 4 KB):
//   [H_D1, +225)  H[k][k-1]            15 x 15, row-major
//   [H_D0, +120)  H[k][k]              LOWER triangle, entry (a, c <= a) at a (a + 1) / 2 + c (the row-per-lane panel of the
//                                      sweep never uses the upper one)
//   [H_D2, +36), [H_D3, +36)           H[k][k-2], H[k][k-3]: pose x pose, 6 x 6
//   [H_DX, +54)   (pose of k) x (velocity / bias of k-2), 6 x 9: filled by the marginal prior for the window's third keyframe
// (the first layout was four 15 x 15 blocks: the solver fetched 5.2 KB of lines per keyframe for the 3.5 KB it needs)
constexpr int HROW = 512;
constexpr int H_D1 = 0, H_D0 = 225, H_D2 = 345, H_D3 = 381, H_DX = 432;
__host__ __device__ constexpr int h_tri(int a, int c) { return a * (a + 1) / 2 + c; }
// Cholesky panel of one keyframe in HBM: 43 rows (27 sub-diagonal rows, the rhs row, 15 rows of L^-T) x 15 columns, stored
// by column PAIRS [pair c][rows_c][2] + column 14 alone [43].  Row r' of L^-T (panel row 28 + r') is zero in front of its
// diagonal, so pair c (columns 2c, 2c+1) keeps only the rows 0 .. 29 + 2c: 547 doubles instead of 645.  A lane whose entry of
// a pair is such a structural zero stores it to / loads it from the 16-byte cell PANEL_DUMP of the same keyframe (only
// zeros are ever written there), so neither sweep needs a predicate.
__host__ __device__ constexpr int panel_rows(int c) { return 30 + 2 * c; }              // rows kept by column pair c < 7
__host__ __device__ constexpr int panel_off(int c) { return 58 * c + 2 * c * c; }       // its first double; c = 7: column 14
constexpr int PANEL_LAST = panel_off(7);     // 504: column 14, [43]
constexpr int PANEL_DUMP = 548;              // the zero cell (2 doubles)
constexpr int PANEL = 550;
static_assert(PANEL_LAST + 43 <= PANEL_DUMP && PANEL_DUMP % 2 == 0 && PANEL_DUMP + 2 <= PANEL && PANEL % 2 == 0, "panel layout");
// index of entry (row, col) of a keyframe's panel; structural zeros map to the zero cell
__host__ __device__ constexpr int panel_idx(int row, int col) {
    return col >= 14 ? PANEL_LAST + row : (row < panel_rows(col >> 1) ? panel_off(col >> 1) + 2 * row + (col & 1) : PANEL_DUMP);
}
constexpr int SEP = 27;
constexpr int SEPM = SEP * 28;  // 27x27 block + right-hand side column
// separator blocks of one (chunk, window): [sepR 27x28 | sepS 27x28 | sepC 27x27] packed in one slot (2241 doubles, padded
// to a multiple of 8), chunk-major [P][B][SEPK]: the chunks of one rank of a time-sharded window are ONE contiguous slice
// of ONE buffer, i.e. one all-gather per LM trial
constexpr int SEPK = (2 * SEPM + SEP * SEP + 7) / 8 * 8;
constexpr int SEPL = SEP * 64;  // factor of one separator elimination, column-major [27][64]: L (27 rows), Z (27), y
constexpr int VROW = 15 * 32;   // spike rows of one keyframe: 15 dof x 27 separator columns (32 stored)

// chunk geometry of an n-keyframe window cut into (at most) P chunks: the chunks that are followed by a separator
// have L or L + 4 pivots (multiples of 4: the sweep's unroll), one cut keyframe sits between consecutive chunks,
// the last chunk takes the rest:  n = sum_{c < P-1} (pivots_c + 1) + last,  with the first `longer` chunks at L + 4
// so that the last one is no longer than L + 4 either
struct ChunkPlan { int L, longer; };
__host__ __device__ inline ChunkPlan chunk_plan(int n, int P) {
    ChunkPlan p{0, 0};
    const int total = n - (P - 1);                     // pivots of all chunks
    if (P < 1 || total < 8 * P) { p.L = total > 0 && P >= 1 ? (total / P) & ~3 : 0; return p; }
    p.L = (total / P) & ~3;
    const int last0 = n - (P - 1) * (p.L + 1);         // the last chunk if no chunk were longer
    int x = (last0 - (p.L + 4) + 3) / 4;
    if (x < 0) x = 0;
    if (x > P - 1) x = P - 1;
    p.longer = x;
    return p;
}
__host__ __device__ inline int chunk_len(int n, int P) { return chunk_plan(n, P).L; }
// chunks actually used for an n-keyframe window: at most P; with `fit` also at most sqrt(n), the
// measured optimum of (chunk sweeps ~ 3.7 us * n / P) + (two-sided separator chain, see DESIGN.md "K4p")
__host__ __device__ inline int chunk_count(int n, int P, int fit) {
    if (fit) {
        int want = 1;
        while ((want + 1) * (want + 1) <= n) want++;
        if (P > want) P = want;
    }
    while (P > 1 && chunk_len(n, P) < 8) P--;
    return P < 1 ? 1 : P;
}
// i0 = first pivot keyframe (window-local), ni = pivots, has_sep: the cut keyframe i0 + ni and the pose rows of the
// two keyframes after it follow as this chunk's tail rows (the next chunk starts at i0 + ni + 1)
struct ChunkGeom { int i0, ni, has_sep; };
__host__ __device__ inline ChunkGeom chunk_geom(int n, int Pe, int c) {
    const ChunkPlan p = chunk_plan(n, Pe);
    ChunkGeom g;
    g.i0 = c * (p.L + 1) + 4 * (c < p.longer ? c : p.longer);
    g.has_sep = c < Pe - 1;
    g.ni = g.has_sep ? p.L + (c < p.longer ? 4 : 0) : n - g.i0;
    return g;
}

// Device-resident problem: B windows x M keyframe slots (G = B*M).  See DESIGN.md.
struct View {
    int B, M;
    long G;
    double grav[3];
    double* x;          // [2][16][G]          states, SoA, double-buffered (current / trial)
    double* imu_in;     // [G/64][190][64]     AoSoA
    double* imu_r;      // [2][G/64][15][64]   whitened residuals, AoSoA, double-buffered (current / LM trial)
    double* imu_j;      // [2][G/8][2496]      J stream (see IMU_OUT above), double-buffered
    int* btw_a;         // [G]                 local index of key a (b = slot), -1 = empty
    double* btw_in;     // [G/64][28][64]
    double* btw_out;    // [2][G/64][78][64]
    int* prior_k;       // [B]                 local keyframe index, -1 = none
    double* prior_in;   // [B][31]
    double* prior_out;  // [2][B][240]
    // marginal prior left by fixed-lag marginalisation: Gaussian on [lo: 15][lo+1: pose 6][lo+2: pose 6]
    int* mp_on;         // [B]
    double* mp_x;       // [B][3][16]   linearisation states
    double* mp_L;       // [B][27][27]  information
    double* mp_eta;     // [B][27]      gradient at the linearisation point
    double* mp_out;     // [2][B][28]   L d + eta (27) and the cost 0.5 d^T L d + eta^T d
    double* H;          // [G][512]            block row of keyframe k ("Block row of H" above)
    double* gvec;       // [G][15]
    double* zrow;       // [900] zeros
    double* delta;      // [G][15] + [B]       increments; tail: solve-failure flag per window (time-sharded windows: summed
                        //                     over the ranks together with the increments, one all-reduce)
    double* Lp;         // [G][PANEL = 550]    Cholesky panels packed to their profile ("Cholesky panel of one keyframe" above: 547 of the 43 x 15
                        //                     doubles; vf_engine_read_panels hands out the documented [43][16])
    // partitioned solve (allocated when P >= 2)
    int P;              // chunks per window (0/1 = whole-window sweeps)
    int P_fit;          // 1: use fewer chunks on short windows (see chunk_count)
    // time-sharded windows (multi-GPU): rank sh_r of sh_G owns the chunks [sh_r Pe / sh_G, (sh_r+1) Pe / sh_G)
    // of every window; states and factors are replicated, each rank linearises / assembles / eliminates its own
    // keyframes only.  sh_G <= 1: not sharded.
    int sh_r, sh_G;
    int sh_all_jac;     // != 0: a time-sharded rank writes the Jacobian of EVERY factor, not only of those that feed its rows (the
                        // refined solve applies J as an operator on whole increments, replicated on every rank)
    double* Vp;         // [G][15][32]         spikes: L^-1 (coupling of the chunk interior to its left separator)
    // the three below point into ONE buffer [P][B][SEPK] (at offsets 0, SEPM, 2 SEPM of a slot): element (c, w) of each sits
    // ((size_t)c * B + w) * SEPK further on
    double* sepR;       // [27][28]            separator block + rhs left by the forward sweep of chunk c
    double* sepS;       // [27][28]            Schur term of chunk c on its LEFT separator (c >= 1)
    double* sepC;       // [27][27]            coupling (right separator of chunk c) x (left separator of chunk c)
    double* sepL;       // [B][P][27][64]      factors of the separator chain, column-major (for its back substitution)
    int* lo;            // [B] active range [lo, hi)
    int* hi;
    int* sel;           // [B] which buffer is current
    int* fail;          // [B] solve failure flag of the current trial
    int* fresh;         // [B] 1 = the current linearisation changed since H, g were last assembled
    double* lambda;     // [B]
    double* cost;       // [B]
    int* n_acc;         // [B]
    int* n_rej;
    int* n_fail;
    double lam_up, lam_down, lam_min, lam_max;
    double accept_rel;  // an LM trial is accepted iff new cost < cost + accept_rel * cost (vf_engine_opts.accept_rel)
    // ... or, with min_fidelity > 0 (vf_engine_opts.min_model_fidelity), by GTSAM's rule: (cost - new cost) / model[w] > min_fidelity,
    // model[w] = the decrease the linearised problem predicts for the step, 0.5 (lambda |delta|^2 - g . delta) (k_model_change)
    double min_fidelity;
    double* model;      // [B]
    // Non-monotone LM (vf_engine_opts.lm_excursion = nm_W > 0): up to nm_W consecutive trials that RAISE the cost are kept
    // provisionally (an "excursion"); the first one saves the point it left (x_best, ref_cost).  A later trial whose cost is
    // below ref_cost ends the excursion with everything accepted; the nm_W + 1-th that is not restores x_best (relin[w] = 1:
    // its factors are linearised again by the conditional launch that follows k_decide).
    int nm_W;
    double* x_best;     // [16][G]
    double* ref_cost;   // [B]
    int* prov;          // [B] provisional trials since the reference point (0: the current point is the reference)
    int* n_prov;        // [B] provisional trials over the life of the engine
    int* relin;         // [B] the current buffer was restored: linearise it again
    int* carry;         // [B] the last solve ended inside an excursion: the next one keeps its lambda (k_reset_lambda)
    int relin_only;     // the linearisation kernels skip windows whose relin flag is clear
    double gauge_floor; // k_marginalize: eigenvalues of the marginal prior's information about the window's global translation and yaw
                        // that have decayed below this are lifted back to it (vf_engine_opts.gauge_floor; 0 = off)
    int w_first;        // K3 only: the first window of its grid (launch_assemble_window: H and g of ONE window on demand)
    // optional LM termination (off by default: every vf_engine_iterate runs its fixed number of trials).  With
    // stop_on, a window whose trial changes the cost by <= abs_tol, or by <= rel_tol relative to the cost
    // (gtsam::LevenbergMarquardtParams relativeErrorTol / absoluteErrorTol, checkConvergence; applied to
    // rejected trials as well), is done: every kernel of the remaining trials skips it.
    int stop_on;
    double rel_tol, abs_tol;
    int* done;          // [B]
    // Hybrid K4 under the termination rule (batches of whole-window sweeps): the sweep's launch time is ONE window's
    // dependency chain whatever the number of windows still taking trials, so once few are left the partitioned form is
    // faster (1.3 ms for 128 windows against 3.1).  Both forms are launched; each kernel looks at n_active and one of the
    // two returns at once (no host synchronisation inside a solve).  gate: 0 = off, 1 = this is the sweep (runs when more
    // than gate_T windows are active), 2 = this is the partitioned form (runs otherwise).
    int* n_active;      // [1]  windows of the current solve that still take trials (k_count_active)
    int gate, gate_T;
    int* act;           // [B] or null: compacted list of the windows still taking trials (k_count_active); the one-wave sweeps of a hybrid
                        // solve then map workgroup i to window act[i], so that the active windows are dispatched first and contiguously
    int tw_max;         // whole-window sweeps: up to this many windows two waves per window from both ends, above one wave per window
    int split_min;      // whole-window sweeps: from this many windows on, forward sweep and back substitution as two kernels (0 = never)
    int asm_min;        // whole-window sweeps: from this many windows on, the forward sweep assembles H itself from the J stream and
                        // K3 is not launched (0 = never; asm_in_solve() below)
    int asm_waves;      // ... as one wave per window (1) or as an eliminator wave + an assembler wave sharing the window's LDS (2)
    unsigned* place;    // [PLACE_CELLS] per-CU claims of the two-wave sweep's launch (which SIMDs have their eliminator): k_band_forward_asm2
    // "far" between factors: BetweenFactor<Pose3> on any pair of keyframes of a window (wider than the band, or a second factor
    // on an end key), at most x_max per window; kept out of the banded H and solved as a low-rank correction
    int x_max;          // 0 until vf_engine_set_extra_between is first called
    int* x_a;           // [B][x_max] window-local slots of the two keyframes, -1 = empty
    int* x_b;
    double* x_in;       // [B][x_max][28]
    double* x_out;      // [2][B][x_max][78] linearisations, double-buffered like btw_out
    // LINEAR far factors: what far factors become when the marginalisation eliminates their older keyframe (k_marginalize).
    // Marginalising a keyframe that several far factors touch couples their far ends, so the window holds them as ONE linear
    // factor: xl_n far ends, six whitened rows per far end, every row over [lo: 15][lo+1: pose][lo+2: pose][far end 0: pose] ..
    // [far end xl_n-1: pose] (row stride xl_ld(v) = 27 + 6 x_max), and a residual at the linearisation point -- the marginal prior's own (mp_x)
    // for the three head keyframes, xl_bx for the far ends:
    //     r(x) = r0 + U [Local(mp_x -> x_lo .. x_lo+2) (27); Local(xl_bx[e] -> x_{b_e}) (6 each)]
    // Re-expressed by every later marginalisation (its support always includes the keyframe that leaves); a far end that
    // comes within the prior's reach is folded into the prior.  In a window's slot numbering the far ends come first: slot
    // s < xl_n is rows 6 s .. 6 s + 5, slot xl_n + i is entry i of x_a / x_b.  Linear + nonlinear <= x_max per window.
    int* xl_n;          // [B]
    int* xl_b;          // [B][x_max] window-local keyframe of each far end
    double* xl_U;       // [B][6 x_max][27 + 6 x_max]
    double* xl_r0;      // [B][6 x_max]
    double* xl_bx;      // [B][x_max][7] (q w x y z, t)
    double* xl_out;     // [2][B][6 x_max] residual at the states of each buffer
    double* far_scratch; // [B][FAR_SCRATCH] (engines made for more than MAX_EXTRA far factors per window) or [B][64]
    int far_big;         // the device-memory forms whatever the slots in use (vf_engine_tuning.far_big_forms; such engines only)
    // Incremental Gauss-Newton updates (vf_engine_opts.incremental; csrc "suffix re-elimination"): the banded form of what
    // ISAM2::update does with relinearizeSkip 1 (GraphManager.cpp:37-43,126-127).  The forward sweep is causal in time, so the
    // panel of a keyframe depends on nothing newer than the factors that touch it: after an update only the keyframes from
    // the first one whose linearisation point moved, or whose factors changed, are linearised, assembled and eliminated
    // again; the sweep restarts from a CHECKPOINT of its trailing window -- the partially eliminated 27 x 27 block (+ rhs) of
    // [k: 15][k+1: pose][k+2: pose] that the elimination of the keyframes in front of k leaves (the sep_out of a chunk sweep) --
    // kept for every CK-th keyframe slot; and the back substitution stops once three consecutive increments come out as
    // they were (the wildfire threshold of iSAM2; 0 = to the bit).
    int inc_on;         // 0: off.  != 0: K1 / K2 / K3 work from inc_k[w] on (0 elsewhere in the library)
    int inc_prior;      // != 0: the priors are linearised whatever inc_k says (the marginal prior has just changed: a slide)
    int* inc_k;         // [B] first keyframe slot changed since the factorisation in Lp / ck was made (INT_MAX: none)
    int* inc_stop;      // [B] slot at which the last back substitution stopped: increments below it are as they were
    int* inc_from;      // [B] slot at which the last forward sweep started (diagnostics: vf_engine_incremental_info)
    double* ck;         // [G / CK][CK_SZ] checkpoints (slot k -> entry k / CK of its window, written by every sweep that passes it)
    double wildfire;    // back substitution: an increment that changes by at most this in every component counts as unchanged
};
__host__ __device__ inline int xl_ld(const View& v) { return 27 + 6 * v.x_max; }   // row stride of View::xl_U
constexpr int CK_LOG = 3, CK = 1 << CK_LOG;       // a checkpoint every 8 keyframe slots
constexpr int CK_SZ = 768;                        // 27 x 28 doubles (SEPM), padded
// first keyframe slot the incremental forward sweep eliminates again, given the first changed slot: a factor reaches three
// keyframes back (the rows of H from inc_k - 3 on are new), and the checkpoint at m holds the rows m .. m + 2 at their values
// of then, so m + 2 < inc_k - 3; m on the checkpoint grid; a sweep that would start within the window's first rows (where
// the prior / the marginal prior sit) starts at the window's first keyframe instead, from nothing
__host__ __device__ inline int inc_start(int inc_k, int lo) {
    if (inc_k < lo + 6) return lo;
    const int m = (inc_k - 6) & ~(CK - 1);
    return m < lo + 4 ? lo : m;
}

// Work vectors of the refined solve (vf_refine.hip): conjugate gradients on (J^T J + lambda I) d = -J^T r with the operator
// applied through J and the engine's Cholesky solve as the preconditioner.  Allocated on first use.
struct Refine {
    double* x;          // [G][15]            the iterate (starts as the plain normal-equation solution)
    double* p;          // [G][15]            search direction
    double* Ap;         // [G][15]            J^T (J p) + lambda p
    double* nres;       // [G][15] + 64       MINUS the residual -g - A x: the right-hand side the correction solves take as "g"
    double* z;          // [G][15] + [B]      M^-1 res, shaped like View::delta (failure flags behind it on time-sharded engines)
    double* u_imu;      // [15][G]            J p, rows of the IMU factor in each slot
    double* u_btw;      // [6][G]             ... of the between factor in each slot
    double* u_pri;      // [B][15]            ... of the prior
    double* rz;         // [B]                res . M^-1 res (< 0: no direction yet)
    double* rz0;        // [B]                its first value
    int* stop;          // [B]                the window takes no further part (converged, failed, empty)
    int* iters;         // [B]                corrections applied in the last solve
    double* part;       // [B][64]            partial sums of the two dot products of a correction (k_pcg_dot)
    double* coef;       // [B]                beta / alpha of the step in flight (k_pcg_scalar -> k_pcg_axpy)
    int* first;         // [B]                the direction in flight is the first one (p := z)
};
void launch_refine_begin(const View& v, const Refine& q, hipStream_t s);                  // x := delta, nres := g + A x
void launch_refine_step(const View& v, const Refine& q, double rel_stop, hipStream_t s);  // after z := M^-1 res: direction, A p, update
void launch_refine_end(const View& v, const Refine& q, hipStream_t s);                    // delta := x
void launch_refine_apply(const View& v, const Refine& q, const double* p, double* out, hipStream_t s);   // out := J^T (J p) + lambda p

// does launch_band_solve(v) assemble the normal equations inside the forward sweep (so that launch_assemble may be skipped)?
// One-wave whole-window sweeps of an unsharded engine only; the far-factor correction and the hybrid form solve from H.
inline bool asm_in_solve(const View& v) {
    return v.asm_min > 0 && v.B >= v.asm_min && v.P < 2 && v.B > v.tw_max && v.sh_G <= 1 && v.gate == 0;
}
// the same question for the SWEEP half of a hybrid solve (launch_band_solve_hybrid; K3 then serves the partitioned half only)
inline bool asm_in_hybrid(const View& v) {
    return v.asm_min > 0 && v.B >= v.asm_min && v.P < 2 && v.B > v.tw_max && v.sh_G <= 1;
}
// isotropic IMU covariances (ImuManagerRos.cpp:20-33)
struct ImuCov { double acc, gyro, integration, bias_acc, bias_omega, bias_int; };
void launch_preintegrate(const View& v, long g0, int n, const int* off, const double* steps, const double* bhat6,
                         const ImuCov& prm, int* status, hipStream_t s);
// the ingest half of a fixed-lag update, all windows in one launch: K0 of the factor ending at slot hi[w] with the window's
// current bias estimate + the between record ending there (off: [B + 1] step offsets, tail_a: [B], tail_btw: [B][28])
void launch_ingest_tail(const View& v, const int* off, const double* steps, const int* tail_a, const double* tail_btw,
                        const ImuCov& prm, int* status, hipStream_t s);
void launch_linearize_extra(const View& v, int which, hipStream_t s);       // far between factors (no-op while x_max == 0)
void launch_extra_gradient(const View& v, hipStream_t s);                    // g += J^T r of the far factors, after K3
void launch_extra_rhs(const View& v, int slot, int row, double* gtmp, hipStream_t s);
void launch_extra_combine(const View& v, const double* Zm, size_t zstride, int slots, hipStream_t s);
void launch_cols_prepare(const View& v, const View& c, int ncols, hipStream_t s);    // single-window engines: the Woodbury columns as one batch
void launch_cols_fail(const View& v, const View& c, int ncols, hipStream_t s);
void launch_partitioned_solve(const View& v, hipStream_t s);
void launch_linearize_imu(const View& v, int which, hipStream_t s);
void launch_linearize_between(const View& v, int which, hipStream_t s);
void launch_linearize_prior(const View& v, int which, hipStream_t s);
void launch_linearize_between_prior(const View& v, int which, hipStream_t s);   // K2 + K2b in one launch (large batches)
void launch_linearize_tail(const View& v, int nslid, hipStream_t s);   // warm start: factors of the appended keyframes + priors
void launch_linearize_all(const View& v, int which, hipStream_t s);   // the three above in one launch (few windows)
void launch_assemble(const View& v, hipStream_t s);
void launch_assemble_window(const View& v, int window, hipStream_t s);   // H and g of one window, whatever its fresh / done flags say
void launch_assemble_for_partitioned(const View& v, hipStream_t s);   // hybrid solves with an assembling sweep (asm_in_hybrid)
void launch_band_solve(const View& v, hipStream_t s);
void launch_count_active(const View& v, hipStream_t s);
// hybrid K4 (see View::gate): vp = the same engine viewed with the partitioned form's chunk count
void launch_band_solve_hybrid(const View& v, const View& vp, hipStream_t s);
void launch_retract(const View& v, hipStream_t s);
void launch_model_change(const View& v, hipStream_t s);     // View::model of every window, from g and the increment just solved
void launch_decide(const View& v, int init, hipStream_t s);
void launch_close_excursions(const View& v, hipStream_t s);
void launch_reset_lambda(const View& v, const double* lambda0, int clear_done, hipStream_t s);   // lambda := lambda0 at the start of a solve (see k_reset_lambda)   // non-monotone LM: undo an excursion left open at the end of a solve
void launch_partitioned_local(const View& v, hipStream_t s);    // chunk sweeps + spikes of the owned chunks
void launch_partitioned_global(const View& v, hipStream_t s);   // separator chain (all of it) + back substitution of the owned chunks
void launch_mask_delta(const View& v, hipStream_t s);           // zero the increments of keyframes this rank does not own
void launch_predict(const View& v, int window, int k0, int n, int from_trial, hipStream_t s);
void launch_relinearize(const View& v, double threshold, hipStream_t s);   // reference-compat solves: theta <- theta (+) delta where |delta| >= threshold
// incremental updates (View::inc_*): relinearise where the pending increment reaches the threshold and note the first slot
// that changed (with `appended` keyframes new at the window's end; invalid: everything), the suffix solve, the estimate
void launch_inc_begin(const View& v, double threshold, int appended, int invalid, hipStream_t s);
void launch_inc_solve(const View& v, hipStream_t s);
void launch_inc_retract(const View& v, hipStream_t s);
void launch_slide(const View& v, const double* sigma15_dev, int reanchor, hipStream_t s);
// host scalars into device state without a copy (asynchronous staging: vf_engine_set_async): the active range of a window
// (lo < 0: keep), every window's first keyframe moved on by one, one between record (28 doubles by value) into the slot of
// its end key, and the result of a solve -- a state, the failure count, the sticky status words (cleared) -- into pinned memory
struct BtwArg { double r[BTW_IN]; };
struct SolveResult { double state[16]; double cost; int n_acc, n_rej, n_fail, sticky[2], pad; };
void launch_set_range(const View& v, int window, int lo, int hi, hipStream_t s);
void launch_bump_lo(const View& v, hipStream_t s);
void launch_put_between(const View& v, long g, int a, const BtwArg& rec, hipStream_t s);
void launch_read_result(const View& v, int window, int slot, int which, int* sticky, SolveResult* out, hipStream_t s);
void launch_marginalize(const View& v, int far_slots_in_use, int* status, hipStream_t s);
constexpr int MARG_STASH_DOUBLES = 729 + 27 + 48 + 4;     // per window: what k_marginalize leaves in a stash (information, gradient, linearisation states)
void launch_marginalize_ahead(const View& v, int* status, double* stash, hipStream_t s);   // engines without far factors
void launch_marg_commit(const View& v, const double* stash, hipStream_t s);
void launch_shift_copy(const double* src, double* dst, long n, hipStream_t s);
void launch_shift_btw_a(int* a, long G, int M, int shift, hipStream_t s);
// AoS <-> AoSoA staging
void launch_scatter(const double* aos, double* aosoa, long g0, long n, int nf, hipStream_t s);
void launch_gather(const double* aosoa, double* aos, long g0, long n, int nf, hipStream_t s);
void launch_scatter_states(const double* aos, double* x, long G, int buf, long g0, long n, hipStream_t s);
// the documented (r | J) records of IMU factors [g0, g0 + n) of buffer b, AoS [n][465], from imu_r / imu_j
void launch_gather_imu_lin(const View& v, int b, long g0, long n, double* aos, hipStream_t s);
void launch_gather_states(const double* x, double* aos, long G, const int* sel, int M, int which, long g0, long n, hipStream_t s);

}  // namespace vf
