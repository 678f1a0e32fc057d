// Host-logic test double for the engine half of libvilfusion (tests only; never shipped): every vf_engine_* entry point
// vf_graph.cpp calls, with no device behind it, so that the GraphManager bookkeeping (queues, give-back, the two-lock
// discipline of GraphManager.h:103-104) can be driven on a CPU and under ThreadSanitizer.
#include <atomic>
#include <cstring>
#include <string>
#include <vector>

#define VF_NO_SIZED_DEFAULTS      /* this file DEFINES the functions the header's macros of the same name would shadow */
#include "../../include/vilfusion.h"

// (the far list is kept the way the real engine keeps it: replaced by vf_engine_set_extra_between, transported to the next
// keyframe when its older keyframe is dropped, shifted by a compaction)
struct FarEntry { int a, b; double rec[VF_BTW_RECORD]; };
struct vf_engine { int lo = 0, hi = 0; std::vector<FarEntry> far; };
static thread_local std::string g_err;
std::atomic<int> fake_fail_preintegrate{0};   // != 0: vf_engine_preintegrate fails (the solve gives its queues back)
std::atomic<long> fake_iterates{0};
// what the last vf_engine_set_between / vf_engine_set_extra_between calls carried (host-logic tests of the far-factor routing)
std::atomic<int> fake_band_n{-1}, fake_extra_n{-1}, fake_extra_calls{0}, fake_extra_a0{-1}, fake_extra_b0{-1};

extern "C" {
const char* vf_last_error(void) { return g_err.c_str(); }
void vf_set_last_error_(const char* m) { g_err = m ? m : ""; }
void vf_engine_default_opts(vf_engine_opts* o) { memset(o, 0, sizeof(*o)); o->struct_size = (uint32_t)sizeof(*o); o->windows = 1; o->capacity = 1088; o->bandwidth = 3; }
int vf_engine_incremental_info(vf_engine*, int, long* u, long* f, int* a, int* b) { if (u) *u = 0; if (f) *f = 0; if (a) *a = -1; if (b) *b = -1; return VF_OK; }
int vf_engine_create(const vf_engine_opts*, vf_engine** out) { *out = new vf_engine(); return VF_OK; }
void vf_engine_destroy(vf_engine* e) { delete e; }
int vf_engine_set_states(vf_engine*, int, int, int, const double*) { return VF_OK; }
int vf_engine_set_prior(vf_engine*, int, int, const double*) { return VF_OK; }
int vf_engine_set_range(vf_engine* e, int, int lo, int hi) { e->lo = lo; e->hi = hi; return VF_OK; }
int vf_engine_set_convergence(vf_engine*, double, double) { return VF_OK; }
int vf_engine_set_imu(vf_engine*, int, int, int, const double*) { return VF_OK; }
int vf_engine_preintegrate(vf_engine*, int, int, int, const int32_t*, const double*, const double*, const vf_imu_params*) {
    if (fake_fail_preintegrate.load()) { g_err = "fake: preintegration refused"; return VF_ERR_DEVICE; }
    return VF_OK;
}
int vf_engine_predict(vf_engine*, int, int, int) { return VF_OK; }
int vf_engine_predict_from_estimate(vf_engine*, int, int, int) { return VF_OK; }
int vf_engine_set_between(vf_engine*, int, int n, const int32_t*, const int32_t*, const double*) { fake_band_n = n; return VF_OK; }
int vf_engine_set_extra_between(vf_engine* e, int, int n, const int32_t* a, const int32_t* b, const double* rec) {
    e->far.clear();
    for (int i = 0; i < n; i++) { FarEntry f; f.a = a[i]; f.b = b[i]; memcpy(f.rec, rec + (size_t)i * VF_BTW_RECORD, sizeof(f.rec)); e->far.push_back(f); }
    fake_extra_n = n;
    fake_extra_calls++;
    fake_extra_a0 = n ? a[0] : -1;
    fake_extra_b0 = n ? b[0] : -1;
    return VF_OK;
}
int vf_engine_marginalize(vf_engine*) { return VF_OK; }
int vf_engine_refine_count(vf_engine*, int* n) { if (n) *n = 0; return VF_OK; }
int vf_engine_read_excursions(vf_engine*, int, int* a, int* b) { if (a) *a = 0; if (b) *b = 0; return VF_OK; }
int vf_engine_drop_oldest(vf_engine* e) {
    std::vector<FarEntry> keep;
    for (auto f : e->far) {
        if (f.a == e->lo) { if (f.b - f.a <= 3) continue; f.a++; }      // (short enough for the marginal prior: absorbed)
        keep.push_back(f);
    }
    e->far.swap(keep);
    e->lo++;
    return VF_OK;
}
int vf_engine_compact(vf_engine* e, int shift) {
    for (auto& f : e->far) { f.a -= shift; f.b -= shift; }
    e->lo -= shift; e->hi -= shift;
    return VF_OK;
}
int vf_engine_get_extra_between(vf_engine* e, int, int* n, int32_t* a, int32_t* b, double* rec, long* tr, long* en, long* ab) {
    if (n) *n = (int)e->far.size();
    for (size_t i = 0; i < e->far.size(); i++) {
        if (a) a[i] = e->far[i].a;
        if (b) b[i] = e->far[i].b;
        if (rec) memcpy(rec + i * VF_BTW_RECORD, e->far[i].rec, sizeof(e->far[i].rec));
    }
    if (tr) *tr = 0;
    if (en) *en = 0;
    if (ab) *ab = 0;
    return VF_OK;
}
int vf_engine_get_linear_far(vf_engine*, int, int* n, int32_t*) { if (n) *n = 0; return VF_OK; }
int vf_engine_grow(vf_engine*, int) { return VF_OK; }
int vf_engine_isam_step(vf_engine*, double) { return VF_OK; }
int vf_engine_iterate(vf_engine*, int) { fake_iterates++; return VF_OK; }
int vf_engine_read_lm(vf_engine*, int, double* c, double* l, int* a, int* r, int* f) {
    if (c) *c = 0; if (l) *l = 0; if (a) *a = 0; if (r) *r = 0; if (f) *f = 0;
    return VF_OK;
}
static int fill_state(int n, double* s) { for (int i = 0; i < n; i++) { memset(s + 16 * i, 0, 16 * sizeof(double)); s[16 * i] = 1.0; } return VF_OK; }
int vf_engine_get_states(vf_engine*, int, int, int n, double* s) { return fill_state(n, s); }
int vf_engine_get_estimate(vf_engine*, int, int, int n, double* s) { return fill_state(n, s); }
int vf_engine_set_async(vf_engine*, int) { return VF_OK; }
int vf_engine_marginalize_ahead(vf_engine*) { return VF_OK; }
int vf_engine_read_result(vf_engine*, int, int, int, double* s, double* c, int* a, int* r, int* f, int* flags) {
    if (c) *c = 0.0;
    if (a) *a = 0;
    if (r) *r = 0;
    if (f) *f = 0;
    if (flags) *flags = 0;
    return s ? fill_state(1, s) : VF_OK;
}
int vf_engine_get_imu(vf_engine*, int, int, int n, double* r) { memset(r, 0, sizeof(double) * VF_IMU_RECORD * n); return VF_OK; }
}
