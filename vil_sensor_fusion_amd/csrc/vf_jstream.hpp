// vf_jstream.hpp -- the order of the J stream (vf_kernels.hpp "imu_j"): shared by the kernels that write it (K1), that
// copy it into LDS (K3, the assembling sweeps) and that apply J as an operator (vf_refine.hip).
#pragma once
#include "vf_kernels.hpp"

namespace vf {

// per-keyframe tangent component c (0..14: theta, p, v, bias_acc, bias_gyro) of side i / j -> column of the 15x30 Jacobian in
// GTSAM key order X_i(6) V_i(3) X_j(6) V_j(3) B_i(6) B_j(6)
__host__ __device__ constexpr int imu_col(int side_j, int c) {
    return c < 9 ? c + (side_j ? 9 : 0) : c + 9 + (side_j ? 6 : 0);
}
// Order in which K1 produces the non-zero entries of the whitened 15x30 Jacobian (column index = GTSAM key order
// X_i(6) V_i(3) X_j(6) V_j(3) B_i(6) B_j(6)), split by the keyframe the column belongs to: this IS the order of the J
// stream in HBM (vf_kernels.hpp "imu_j"), so K1 stores pairs of consecutive entries of a side as it goes.
struct JMap {
    short idx[450];          // entry (row * 30 + col) -> position in its side's stream, -1 = structural zero
    short fld[2 * JS_PAIRS]; // stream word (2 * pair + half, i-side pairs first) -> row * 30 + col, -1 = padding
    int n[2];
};
__host__ __device__ constexpr bool jcol_is_j(int col) { return (col >= 9 && col < 18) || col >= 24; }
constexpr JMap make_jmap() {
    JMap m{};
    for (int i = 0; i < 450; i++) m.idx[i] = -1;
    for (int i = 0; i < 2 * JS_PAIRS; i++) m.fld[i] = -1;
    m.n[0] = m.n[1] = 0;
    auto emit = [&m](int row, int col) {
        const int side = jcol_is_j(col) ? 1 : 0;
        const int e = m.n[side]++;
        m.idx[row * 30 + col] = (short)e;
        m.fld[(side ? 2 * JS_PI : 0) + e] = (short)(row * 30 + col);
    };
    for (int c = 0; c < 3; c++) {               // (the loop nest of linearize_imu_core)
        for (int a = 0; a < 9; a++) emit(a, c);
        for (int a = 0; a < 6; a++) emit(a, 3 + c);
        for (int a = 0; a < 9; a++) emit(a, 6 + c);
        for (int a = 0; a < 9; a++) emit(a, 9 + c);
        for (int a = 0; a <= 3 + c; a++) emit(a, 12 + c);
        for (int a = 0; a < 9; a++) emit(a, 15 + c);
    }
    for (int c = 0; c < 6; c++)
        for (int a = 0; a < 10 + c; a++) { emit(a, 18 + c); emit(a, 24 + c); }
    return m;
}
constexpr JMap JM = make_jmap();
static_assert(JM.n[0] == JS_NI && JM.n[1] == JS_NJ, "J stream sizes");
__device__ constexpr JMap JMD = make_jmap();             // the same tables in device memory, for run-time indexed reads

}  // namespace vf
