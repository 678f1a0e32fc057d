// vf_kernels.hpp -- device view of an engine and the kernel launch interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vf {

constexpr int TILE = 64;        // AoSoA tile = one wavefront of factors
constexpr int IMU_IN = 190;     // dt, delta(9), bhat(6), H(54), R packed upper (120)
constexpr int IMU_OUT = 465;    // r(15), J(15x30 row-major)
constexpr int BTW_IN = 28;      // q(4), t(3), R packed upper (21)
constexpr int BTW_OUT = 78;     // r(6), Ja(36), Jb(36)
constexpr int PRIOR_IN = 31;    // mean state (16), sigma (15)
constexpr int PRIOR_OUT = 240;  // r(15), J(15x15)
constexpr int NBLK = 4;         // band blocks stored per keyframe (bandwidth 3 + diagonal)
constexpr int HROW = NBLK * 225;
constexpr int PANEL = 43 * 16;  // Cholesky panel in HBM: 43 rows (27 sub-diagonal, rhs, 15 of L^-T), one 128-B line each
// partitioned solve (K4p): a window is cut into P chunks separated by 3-keyframe (45-dof) separators
constexpr int SEP = 45;
constexpr int SEPM = SEP * 46;  // 45x45 block + right-hand side column
constexpr int SEPL = 45 * 96;   // factor of one separator elimination, column-major [45][96]: L (45 rows), Z (45), y
constexpr int VROW = 15 * 48;   // spike rows of one keyframe: 15 dof x 45 separator columns (48 stored)

// chunk geometry of an n-keyframe window cut into (at most) P chunks: interiors of L keyframes
// (L a multiple of 4), 3 separator keyframes between consecutive chunks, the last chunk takes the rest
__host__ __device__ inline int chunk_len(int n, int P) {
    const int total = n - 3 * (P - 1);                 // interior keyframes
    if (total < 8 * P) return total > 0 ? (total / P) & ~3 : 0;
    const int up = ((total + P - 1) / P + 3) & ~3;     // round up: the last chunk gets the (smaller) rest
    return n - (P - 1) * (up + 3) >= 8 ? up : (total / P) & ~3;
}
// chunks actually used for an n-keyframe window: at most P; with `fit` also at most sqrt(0.32 n), the
// measured optimum of (chunk sweeps ~ 3.7 us * n / P) + (two-sided separator chain ~ 17 us * P / 2 + middle)
__host__ __device__ inline int chunk_count(int n, int P, int fit) {
    if (fit) {
        int want = 1;
        while ((want + 1) * (want + 1) * 100 <= n * 32) want++;
        if (P > want) P = want;
    }
    while (P > 1 && chunk_len(n, P) < 8) P--;
    return P < 1 ? 1 : P;
}
struct ChunkGeom { int i0, ni, has_sep; };   // first interior keyframe (window-local), interior count, separator follows
__host__ __device__ inline ChunkGeom chunk_geom(int n, int Pe, int c) {
    const int L = chunk_len(n, Pe);
    ChunkGeom g;
    g.i0 = c * (L + 3);
    g.has_sep = c < Pe - 1;
    g.ni = g.has_sep ? L : n - g.i0;
    return g;
}

// Device-resident problem: B windows x M keyframe slots (G = B*M).  See DESIGN.md.
struct View {
    int B, M;
    long G;
    double grav[3];
    double* x;          // [2][16][G]          states, SoA, double-buffered (current / trial)
    double* imu_in;     // [G/64][190][64]     AoSoA
    double* imu_out;    // [2][G/64][465][64]  AoSoA, double-buffered
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
    double* H;          // [G][4][15][15]      block d of row k = H[k][k-d]
    double* gvec;       // [G][15]
    double* delta;      // [G][15]
    double* Lp;         // [G][43][16]         Cholesky panels (15 of 16 columns used)
    // partitioned solve (allocated when P >= 2)
    int P;              // chunks per window (0/1 = whole-window sweeps)
    int P_fit;          // 1: use fewer chunks on short windows (see chunk_count)
    // time-sharded windows (multi-GPU): rank sh_r of sh_G owns the chunks [sh_r Pe / sh_G, (sh_r+1) Pe / sh_G)
    // of every window; states and factors are replicated, each rank linearises / assembles / eliminates its own
    // keyframes only.  sh_G <= 1: not sharded.
    int sh_r, sh_G;
    double* cost_part;  // [2][B]              this rank's share of the cost, and its solve-failure flag (summed over ranks by the host side)
    double* Vp;         // [G][15][48]         spikes: L^-1 (coupling of the chunk interior to its left separator)
    double* sepR;       // [P][B][45][46]      separator block + rhs left by the forward sweep of chunk c
    double* sepS;       // [P][B][45][46]      Schur term of chunk c on its LEFT separator (c >= 1)
    double* sepC;       // [P][B][45][45]      coupling (right separator of chunk c) x (left separator of chunk c)
    double* sepL;       // [B][P][45][96]      factors of the separator chain, column-major (for its back substitution)
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
    // optional LM termination (off by default: every vf_engine_iterate runs its fixed number of trials).  With
    // stop_on, a window whose trial changes the cost by <= abs_tol, or by <= rel_tol relative to the cost
    // (gtsam::LevenbergMarquardtParams relativeErrorTol / absoluteErrorTol, checkConvergence; applied to
    // rejected trials as well), is done: every kernel of the remaining trials skips it.
    int stop_on;
    double rel_tol, abs_tol;
    int* done;          // [B]
};

// isotropic IMU covariances (ImuManagerRos.cpp:20-33)
struct ImuCov { double acc, gyro, integration, bias_acc, bias_omega, bias_int; };
void launch_preintegrate(const View& v, long g0, int n, const int* off, const double* steps, const double* bhat6,
                         const ImuCov& prm, int* status, hipStream_t s);
void launch_linearize_imu(const View& v, int which, hipStream_t s);
void launch_linearize_between(const View& v, int which, hipStream_t s);
void launch_linearize_prior(const View& v, int which, hipStream_t s);
void launch_assemble(const View& v, hipStream_t s);
void launch_band_solve(const View& v, hipStream_t s);
void launch_retract(const View& v, hipStream_t s);
void launch_decide(const View& v, int init, hipStream_t s);
// sharded windows: mode 1 = this rank's share of the cost -> cost_part; mode 2 = accept / reject with cost_part as the total
void launch_decide_mode(const View& v, int init, int mode, hipStream_t s);
void launch_partitioned_local(const View& v, hipStream_t s);    // chunk sweeps + spikes of the owned chunks
void launch_partitioned_global(const View& v, hipStream_t s);   // separator chain (all of it) + back substitution of the owned chunks
void launch_mask_delta(const View& v, hipStream_t s);           // zero the increments of keyframes this rank does not own
void launch_predict(const View& v, int window, int k0, int n, hipStream_t s);
void launch_slide(const View& v, const double* sigma15_dev, int reanchor, hipStream_t s);
void launch_marginalize(const View& v, int* status, hipStream_t s);
void launch_shift_copy(const double* src, double* dst, long n, hipStream_t s);
void launch_shift_btw_a(int* a, long G, int M, int shift, hipStream_t s);
// AoS <-> AoSoA staging
void launch_scatter(const double* aos, double* aosoa, long g0, long n, int nf, hipStream_t s);
void launch_gather(const double* aosoa, double* aos, long g0, long n, int nf, hipStream_t s);
void launch_scatter_states(const double* aos, double* x, long G, int buf, long g0, long n, hipStream_t s);
void launch_gather_states(const double* x, double* aos, long G, const int* sel, int M, int which, long g0, long n, hipStream_t s);

}  // namespace vf
