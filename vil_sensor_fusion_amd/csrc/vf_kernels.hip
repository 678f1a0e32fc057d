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

#include "kernels/k0_preintegrate.inc"
#include "kernels/k12_linearize.inc"
#include "kernels/far.inc"
#include "kernels/k2b_priors.inc"
#include "kernels/k3_assemble.inc"
#include "kernels/k4_band_body.inc"
#include "kernels/k4_band.inc"
#include "kernels/k4p_partitioned.inc"
#include "kernels/k5_lm.inc"
#include "kernels/kmarg.inc"
#include "kernels/kstage.inc"
#include "kernels/launch.inc"
}  // namespace vf
