// vf_far.hpp -- slot resolution of the far between factors (View::x_* and View::xl_*), shared by the kernels that apply
// them (vf_kernels.hip: linearisation, gradient, Woodbury columns; vf_refine.hip: the rows of J they add to the operator).
// Included inside namespace vf after VF_DI is defined.
#pragma once
struct FarRef { int kind, idx, a, kb, nl; };
VF_DI FarRef far_ref(const View& v, int w, int s) {
    FarRef f{-1, 0, 0, 0, 0};
    const int nl = v.xl_n[w], lo = v.lo[w], hi = v.hi[w];
    if (s < nl) {
        if (hi - lo > 3 && v.mp_on[w]) { f.kind = 1; f.idx = s; f.a = lo; f.kb = v.xl_b[w * v.x_max + s]; f.nl = nl; }
    } else if (s - nl < v.x_max) {
        const int i = w * v.x_max + s - nl, a = v.x_a[i], kb = v.x_b[i];
        if (!(a < lo || kb >= hi || a >= kb)) { f.kind = 0; f.idx = s - nl; f.a = a; f.kb = kb; }
    }
    return f;
}
VF_DI int far_cols(const FarRef& f) { return f.kind == 1 ? 27 + 6 * f.nl : 12; }
// column c of a slot's six rows -> (window-local keyframe, dof of its 15)
VF_DI void far_col(const View& v, int w, const FarRef& f, int c, int& k, int& d) {
    if (f.kind == 1) {
        if (c < 15) { k = f.a; d = c; }
        else if (c < 21) { k = f.a + 1; d = c - 15; }
        else if (c < 27) { k = f.a + 2; d = c - 21; }
        else { const int e = (c - 27) / 6; k = v.xl_b[w * v.x_max + e]; d = c - 27 - 6 * e; }
    } else if (c < 6) { k = f.a; d = c; }
    else { k = f.kb; d = c - 6; }
}
// entry (row j, column c) of the whitened Jacobian / residual j, at the states of buffer `buf`
VF_DI double far_jac(const View& v, int w, const FarRef& f, int buf, int j, int c) {
    if (f.kind == 1) return v.xl_U[((size_t)w * 6 * v.x_max + 6 * f.idx + j) * xl_ld(v) + c];
    return v.x_out[(((size_t)buf * v.B + w) * v.x_max + f.idx) * BTW_OUT + 6 + (c < 6 ? 0 : 36) + j * 6 + (c < 6 ? c : c - 6)];
}
VF_DI double far_res(const View& v, int w, const FarRef& f, int buf, int j) {
    if (f.kind == 1) return v.xl_out[((size_t)buf * v.B + w) * 6 * v.x_max + 6 * f.idx + j];
    return v.x_out[(((size_t)buf * v.B + w) * v.x_max + f.idx) * BTW_OUT + j];
}
