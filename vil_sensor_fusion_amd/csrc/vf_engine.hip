// vf_engine.hip -- host side of the batch engine behind include/vilfusion.h.
// Owns the HBM-resident problem (vf::View), stages host AoS records into the AoSoA device
// layout, and sequences the hot-path kernels on one HIP stream.  There is no CPU fallback:
// without a gfx950 device every entry point fails with VF_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <dlfcn.h>

#include "../../include/vilfusion.h"
#include "vf_kernels.hpp"

namespace {
thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(VF_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                  \
    } while (0)
}  // namespace

struct vf_engine;
// HIP's current device is per host thread: every entry point that takes an engine makes the engine's device current for
// the duration of the call (allocations, copies and launches would otherwise go to whatever device the calling thread
// last used -- e.g. a ROS spinner thread driving a GraphManager on device 1) and restores the caller's on return.
struct DeviceGuard {
    int prev = -1, want = -1;
    // overlap_ok: the entry point touches nothing the side stream's work does (see vf_engine::side_open)
    explicit DeviceGuard(const vf_engine* e, bool overlap_ok = false);
    ~DeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct vf_engine {
    vf::View v{};
    vf_engine_opts opts{};
    vf_engine_tuning tune{};
    hipStream_t stream = nullptr;
    bool own_stream = true;   // false once the caller has handed in its own stream (vf_engine_set_stream)
    // vf_engine_opts.use_hip_graph: vf_engine_iterate replays its launch sequence (6 + 9 K kernels / memsets) from a captured
    // hipGraph; re-captured when the trial count or any scalar baked into the kernel arguments changes
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_iters = -1, graph_mode = 0;
    long graph_epoch = -1, epoch = 0;
    bool graph_off = false;
    long graph_replays = 0;          // hipGraphLaunch calls so far; graph_captures: (re-)captures of the sequence
    int graph_captures = 0;
    double* lambda0_dev = nullptr;   // [B] lambda0, the source of the per-solve reset
    void drop_graph() {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
        graph_exec = nullptr;
        graph = nullptr;
        graph_iters = -1;
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<void*> allocs;
    double* stage = nullptr;  // device staging buffer (AoS)
    size_t stage_bytes = 0;
    double* sigma_dev = nullptr;
    double sigma_host[15] = {0};    // what sigma_dev holds (vf_engine_slide uploads only a changed set)
    bool sigma_valid = false;
    int* status_dev = nullptr;
    std::vector<int> h_lo, h_hi;  // host mirror of the active ranges
    // Warm start of vf_engine_iterate: true from the end of a solve until any entry point other than vf_engine_slide
    // touches the engine; `slid` counts the slides since.  See k_linearize_tail.
    bool warm = false;
    int slid = 0;             // keyframes appended since the last solve
    int redo = 0;             // slots in front of the old window end whose factors changed since (see touch())
    bool no_warm = false;     // vf_engine_opts.cold_start: every solve starts cold (tests compare the two)
    // Incremental updates (vf_engine_opts.incremental; vf_kernels.hpp "Incremental Gauss-Newton updates"): inc_valid = the panels
    // and checkpoints on the device are those of the problem as it was after the last vf_engine_isam_step, and everything
    // that changed since is an append -- inc_slid slides, or (one-window engines) writes at or beyond slot inc_first_dirty.
    // Any other entry point that writes (cold()) voids it: the next update eliminates the whole window.
    // Asynchronous staging (vf_engine_set_async; the GraphManager's engine): the staging calls of a vf_solve -- preintegrate,
    // set_between, marginalize, drop_oldest, set_range -- enqueue and return; what the device finds wrong (a preintegrated
    // covariance or a marginalisation pivot that is not positive definite) is OR-ed into two sticky words that
    // vf_engine_read_result hands back with the solve's result, in the one synchronisation a solve needs.  The marginalisation
    // of the keyframe that leaves runs on a second stream beside K0 / prediction / staging of the one that arrives (they
    // touch opposite ends of the window): side_open until the main stream has been made to wait for it (join_side, at
    // every entry point that is not one of those three).
    bool async_on = false, side_open = false;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int* sticky_dev = nullptr;
    vf::SolveResult* res_host = nullptr;
    int ensure_async() {
        if (res_host) return VF_OK;
        HIPCHK(hipStreamCreate(&stream2));
        HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        HIPCHK(hipMalloc((void**)&sticky_dev, 2 * sizeof(int)));
        HIPCHK(hipMemsetAsync(sticky_dev, 0, 2 * sizeof(int), stream));
        HIPCHK(hipHostMalloc((void**)&res_host, sizeof(vf::SolveResult), hipHostMallocDefault));
        return VF_OK;
    }
    int join_side() {
        if (!side_open) return VF_OK;
        side_open = false;
        HIPCHK(hipEventRecord(ev_join, stream2));
        HIPCHK(hipStreamWaitEvent(stream, ev_join, 0));
        return VF_OK;
    }
    bool async_now() const { return async_on && x_used == 0 && v.B == 1 && own_stream; }
    // vf_engine_marginalize_ahead: the marginal prior the next vf_engine_marginalize will need, computed behind the solve that has
    // just ended (the linearisation it reads is final by then) into marg_stash; valid while nothing but appends has happened
    // since and the window's first keyframe is still ahead_lo
    double* marg_stash = nullptr;
    bool ahead_valid = false;
    int ahead_lo = -1;
    long ahead_used = 0, ahead_made = 0;
    bool inc_valid = false;
    int inc_slid = 0, inc_first_dirty = 0x7fffffff;
    long inc_updates = 0, inc_full = 0;      // incremental updates so far; those that eliminated from the window's first keyframe
    // hybrid K4 (vf_kernels.hpp "View::gate"): buffers and chunk count of the partitioned form for a sweep engine, allocated
    // when the termination rule is first switched on
    bool hybrid = false;
    int hybrid_P = 0;
    double *h_Vp = nullptr, *h_sep = nullptr, *h_sepL = nullptr;
    int* act_list = nullptr;   // compacted list of the windows still taking trials (vf_engine_opts.hybrid_active_list): handed to the hybrid's
                               // sweeps and to k_count_active only -- every other launch sees View::act = null
    vf::View partitioned_view() const {
        vf::View p = v;
        p.P = hybrid_P;
        p.P_fit = 1;
        p.Vp = h_Vp;
        p.sepR = h_sep;
        p.sepS = h_sep + vf::SEPM;
        p.sepC = h_sep + 2 * vf::SEPM;
        p.sepL = h_sepL;
        return p;
    }

    // The arrays of an engine are carved out of ONE device allocation made at creation (reserve(): a hipMalloc costs 0.1-0.3 ms
    // whatever its size, and an engine has forty arrays -- creating the 6-window column engine of a first loop closure inside a
    // vf_solve took 13 ms of which the memory itself was the least); what does not fit, or comes later, is allocated on its own.
    char* arena = nullptr;
    size_t arena_size = 0, arena_used = 0;
    int reserve(size_t bytes) {
        void* q = nullptr;
        HIPCHK(hipMalloc(&q, bytes));
        allocs.push_back(q);
        HIPCHK(hipMemsetAsync(q, 0, bytes, stream));
        arena = (char*)q;
        arena_size = bytes;
        arena_used = 0;
        return VF_OK;
    }
    template <typename T>
    int alloc(T** p, size_t n, bool zero = true) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (arena && arena_used + bytes <= arena_size) {
            *p = (T*)(arena + arena_used);           // (zeroed by reserve())
            arena_used += bytes;
            return VF_OK;
        }
        void* q = nullptr;
        HIPCHK(hipMalloc(&q, n * sizeof(T)));
        allocs.push_back(q);
        if (zero) HIPCHK(hipMemsetAsync(q, 0, n * sizeof(T), stream));
        *p = (T*)q;
        return VF_OK;
    }
    // grow-only scratch of vf_engine_preintegrate (the GraphManager calls it once per solve: no hipMalloc / hipFree there)
    void* pre_buf = nullptr;
    size_t pre_bytes = 0;
    int ensure_pre(size_t bytes) {
        if (bytes <= pre_bytes) return VF_OK;
        if (pre_buf) HIPCHK(hipFree(pre_buf));
        pre_buf = nullptr;
        pre_bytes = 0;
        const size_t want = bytes < (1u << 16) ? (1u << 16) : bytes * 2;
        HIPCHK(hipMalloc(&pre_buf, want));
        pre_bytes = want;
        return VF_OK;
    }
    // vf_engine_ingest_tail: pinned host staging (so the one host->device copy of an update is asynchronous), its device
    // twin, the events that time the copy and the kernel, and the sticky status word the kernel reports into
    void* in_host = nullptr;
    void* in_dev = nullptr;
    size_t in_bytes = 0;
    int* in_status = nullptr;
    hipEvent_t in_ev[3] = {nullptr, nullptr, nullptr};
    bool in_pending = false;
    int ensure_ingest(size_t bytes) {
        if (!in_ev[0])
            for (auto& ev : in_ev) HIPCHK(hipEventCreate(&ev));
        if (!in_status) {
            HIPCHK(hipMalloc((void**)&in_status, sizeof(int)));
            HIPCHK(hipMemsetAsync(in_status, 0, sizeof(int), stream));
        }
        if (bytes <= in_bytes) return VF_OK;
        if (in_pending) HIPCHK(hipEventSynchronize(in_ev[1]));
        if (in_host) HIPCHK(hipHostFree(in_host));
        if (in_dev) HIPCHK(hipFree(in_dev));
        in_host = in_dev = nullptr;
        in_bytes = 0;
        const size_t want = bytes * 2;
        HIPCHK(hipHostMalloc(&in_host, want, hipHostMallocDefault));
        HIPCHK(hipMalloc(&in_dev, want));
        in_bytes = want;
        return VF_OK;
    }
    // far between factors (View::x_*): host mirror of how many slots are in use (the low-rank correction solves 6 right-hand
    // sides per slot in use), the right-hand-side scratch and the solved columns Z
    int x_used = 0;
    std::vector<int> h_xn;
    std::vector<std::vector<int>> h_xa, h_xb;          // host copies of every window's list (re-sent after compact / grow)
    std::vector<std::vector<double>> h_xrec;
    double* x_gtmp = nullptr;
    double* x_Z = nullptr;
    size_t x_zstride = 0;
    long far_transported = 0, far_ended = 0, far_absorbed = 0;   // far factors moved on to the next keyframe when theirs left the window /
                                                                 // dropped without a marginalisation / absorbed into the marginal prior
    bool marg_since_drop = false;   // vf_engine_marginalize ran since the last vf_engine_drop_oldest (the GraphManager's call pair)
    int x_zslots = 0;          // slots x_Z holds columns for (6 columns each); grown on demand, never beyond VF_MAX_EXTRA
    // the device lists are allocated once (they stay in `allocs`); a failure half way leaves what exists in place and a
    // later call picks up from there (no second allocation, nothing leaked)
    int *x_la = nullptr, *x_lb = nullptr;
    double *x_li = nullptr, *x_lo = nullptr;
    // Woodbury columns of a single-window engine as one batch: a second engine of 6 x VF_MAX_EXTRA windows and the same
    // capacity, on this engine's stream; window q holds a copy of the window's H and column q of U as right-hand side
    // (k_cols_prepare), one partitioned solve of it returns all of Z.  Made when the first far factor arrives, never inside a
    // solve (a solve may be under stream capture); absent (sequential columns, as on batch engines) if it cannot be had.
    vf_engine* far_columns = nullptr;
    bool is_far_columns = false;
    // linear far factors (View::xl_*; made and kept by k_marginalize): the host mirrors only their number and far ends
    std::vector<std::vector<int>> h_lb;
    int h_ln(int w) const { return h_lb.empty() ? 0 : (int)h_lb[w].size(); }
    void attach_far() {              // after ensure_far: the View sees the slot arrays, the host mirrors exist
        if (v.x_max) return;
        v.x_a = x_la; v.x_b = x_lb; v.x_in = x_li; v.x_out = x_lo;
        v.x_max = VF_MAX_EXTRA;
        h_xn.assign(v.B, 0);
        h_xa.assign(v.B, {});
        h_xb.assign(v.B, {});
        h_xrec.assign(v.B, {});
        h_lb.assign(v.B, {});
    }
    void recount_far() {
        x_used = 0;
        for (int w = 0; w < v.B && v.x_max; w++) x_used = std::max(x_used, h_xn[w] + h_ln(w));
    }
    int ensure_far(int slots) {
        const size_t B = (size_t)v.B, X = VF_MAX_EXTRA;
        int rc;
        if (!x_la) { if ((rc = alloc(&x_la, B * X, false))) return rc; HIPCHK(hipMemsetAsync(x_la, 0xff, B * X * sizeof(int), stream)); }
        if (!x_lb) { if ((rc = alloc(&x_lb, B * X, false))) return rc; HIPCHK(hipMemsetAsync(x_lb, 0xff, B * X * sizeof(int), stream)); }
        if (!x_li && (rc = alloc(&x_li, B * X * vf::BTW_IN))) return rc;
        if (!x_lo && (rc = alloc(&x_lo, 2 * B * X * vf::BTW_OUT))) return rc;
        if ((!v.xl_n && (rc = alloc(&v.xl_n, B))) || (!v.xl_b && (rc = alloc(&v.xl_b, B * X))) || (!v.xl_U && (rc = alloc(&v.xl_U, B * X * 6 * vf::XL_LD))) ||
            (!v.xl_r0 && (rc = alloc(&v.xl_r0, B * X * 6))) || (!v.xl_bx && (rc = alloc(&v.xl_bx, B * X * 7))) || (!v.xl_out && (rc = alloc(&v.xl_out, 2 * B * X * 6)))) return rc;
        x_zstride = (size_t)v.G * 15 + B + 64;
        if (!x_gtmp) HIPCHK(hipMalloc((void**)&x_gtmp, x_zstride * sizeof(double)));
        if (slots > x_zslots) {
            // the solved columns of the low-rank correction: 6 increment-shaped columns per slot IN USE (one far factor in
            // one window of a 1024 x 1088 batch engine is 0.8 GB, not the 6.4 GB that VF_MAX_EXTRA slots would be)
            double* z = nullptr;
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipMalloc((void**)&z, 6 * (size_t)slots * x_zstride * sizeof(double)));
            if (x_Z) (void)hipFree(x_Z);
            x_Z = z;
            x_zslots = slots;
        }
        return VF_OK;
    }
    // refined solve (vf_refine.hip): work vectors, allocated on first use; refine_open: a time-sharded caller is between
    // vf_engine_refine_begin and vf_engine_refine_end, and vf_engine_solve_local / _global work on (nres, z)
    vf::Refine rq{};
    bool rq_ready = false, refine_open = false;
    int* rq_stop_host = nullptr;     // [B] pinned: the stop flags, read back between corrections (vf_engine_solve)
    int ensure_refine() {
        if (rq_ready) return VF_OK;
        const size_t G = (size_t)v.G, B = (size_t)v.B;
        int rc;
        if (!rq_stop_host) HIPCHK(hipHostMalloc((void**)&rq_stop_host, B * sizeof(int), hipHostMallocDefault));
        if ((rc = alloc(&rq.x, G * 15)) || (rc = alloc(&rq.p, G * 15)) || (rc = alloc(&rq.Ap, G * 15)) ||
            (rc = alloc(&rq.nres, G * 15 + 64)) || (rc = alloc(&rq.z, G * 15 + B)) || (rc = alloc(&rq.u_imu, G * 15)) ||
            (rc = alloc(&rq.u_btw, G * 6)) || (rc = alloc(&rq.u_pri, B * 15)) || (rc = alloc(&rq.rz, B)) ||
            (rc = alloc(&rq.rz0, B)) || (rc = alloc(&rq.stop, B)) || (rc = alloc(&rq.iters, B)) ||
            (rc = alloc(&rq.part, B * 64)) || (rc = alloc(&rq.coef, B)) || (rc = alloc(&rq.first, B))) return rc;
        rq_ready = true;
        return VF_OK;
    }
    // corrections per solve: vf_engine_opts.refine_iterations, or (auto) 12 once a window is longer than refine_min_keyframes
    int refine_iters() const {
        if (opts.refine_iterations >= 0) return opts.refine_iterations;
        int longest = 0;
        for (int w = 0; w < v.B; w++) longest = std::max(longest, h_hi[w] - h_lo[w]);
        return longest > opts.refine_min_keyframes ? 12 : 0;
    }
    // ... and how far each solve's corrections are driven (auto mode).  What the normal equations get wrong grows smoothly with
    // the window -- plain Gauss-Newton contracts by 0.1 per update at 1 500 keyframes, 0.3 at 2 000, 0.7 at 3 000 -- so the
    // residual reduction a solve is refined to tightens smoothly too, from 1e-3 where refinement starts to refine_rel_stop at
    // twice that length: 1-2 corrections just above the threshold instead of 4, 5 at 3 000 as before (round 5's switch from
    // none to "until 1e-8" made vf_solve four times dearer between 1 500 and 1 600 keyframes).
    double refine_stop() const {
        if (opts.refine_iterations >= 0 || !(opts.refine_rel_stop < 1e-3)) return opts.refine_rel_stop;
        int longest = 0;
        for (int w = 0; w < v.B; w++) longest = std::max(longest, h_hi[w] - h_lo[w]);
        const double n0 = opts.refine_min_keyframes, n1 = 2.0 * n0;
        if (longest >= n1 || n0 <= 0) return opts.refine_rel_stop;
        const double t = std::max(0.0, (longest - n0) / (n1 - n0));
        return std::pow(10.0, -3.0 + t * (std::log10(opts.refine_rel_stop) + 3.0));
    }
    // non-monotone LM (vf_engine_opts.lm_excursion): provisional trials allowed per excursion; auto = 3 on engines that refine
    int excursion() const { return opts.lm_excursion >= 0 ? opts.lm_excursion : (refine_iters() > 0 ? 3 : 0); }
    int ensure_excursion() {
        if (v.x_best) return VF_OK;
        int rc;
        if ((rc = alloc(&v.x_best, 16 * (size_t)v.G)) || (rc = alloc(&v.ref_cost, (size_t)v.B)) || (rc = alloc(&v.prov, (size_t)v.B)) ||
            (rc = alloc(&v.n_prov, (size_t)v.B)) || (rc = alloc(&v.relin, (size_t)v.B)) || (rc = alloc(&v.carry, (size_t)v.B))) return rc;
        return VF_OK;
    }
    int ensure_stage(size_t bytes) {
        if (bytes <= stage_bytes) return VF_OK;
        if (stage) HIPCHK(hipFree(stage));
        stage = nullptr;
        stage_bytes = 0;
        HIPCHK(hipMalloc((void**)&stage, bytes));
        stage_bytes = bytes;
        return VF_OK;
    }
};

// anything the warm start / the incremental update do not know how to follow: the next solve starts from nothing
static inline void cold(vf_engine* e) {
    e->warm = false;
    e->inc_valid = false;
    e->ahead_valid = false;
}

DeviceGuard::DeviceGuard(const vf_engine* e, bool overlap_ok) {
    if (!e) return;
    if (!overlap_ok && e->side_open) (void)const_cast<vf_engine*>(e)->join_side();
    want = e->opts.device;
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipSetDevice(want); return; }
    if (prev != want) (void)hipSetDevice(want);
}

extern "C" {

const char* vf_last_error(void) { return g_err.c_str(); }
void vf_set_last_error_(const char* msg) { g_err = msg ? msg : ""; }  // used by vf_graph.cpp
const char* vf_version(void) { return "vilfusion-mi355x 0.1 (gfx950, float64)"; }

int vf_device_count(int* count) {
    if (!count) return fail(VF_ERR_INVALID, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(VF_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return VF_OK;
}

#undef vf_engine_default_opts
#undef vf_engine_default_tuning
static void engine_defaults(vf_engine_opts* o) {
    memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t)sizeof(*o);
    o->windows = 1;
    o->capacity = 1088;
    o->bandwidth = 3;
    o->device = 0;
    o->gravity[0] = 0.0; o->gravity[1] = 0.0; o->gravity[2] = -9.81;
    // gtsam::LevenbergMarquardtParams defaults (the optimiser commented out at
    // GraphManager.cpp:128-129): lambdaInitial 1e-5, lambdaFactor 10
    o->lambda0 = 1e-5; o->lambda_up = 10.0; o->lambda_down = 10.0;
    o->lambda_min = 1e-12; o->lambda_max = 1e10;
    o->cold_start = 0;
    o->accept_rel = 1e-9;
    o->refine_iterations = -1;       // auto: windows longer than refine_min_keyframes
    o->refine_min_keyframes = 1536;
    o->refine_rel_stop = 1e-8;
    o->lm_excursion = -1;            // auto: 3 on engines that refine, classical LM otherwise
    o->gauge_floor = 3e-4;
    o->incremental = 0;
    o->wildfire = 0.0;
}
static void tuning_defaults(vf_engine_tuning* t) {
    memset(t, 0, sizeof(*t));
    t->struct_size = (uint32_t)sizeof(*t);
    t->sweep_two_sided_max = 256;
    t->hybrid_threshold = 256;   // partitioned form: 0.0095 ms per window; the sweep: 2.9 ms whatever their number
    t->use_hip_graph = 0;
    t->solve_split_min = 2048;
    t->solve_assemble_min = 768;
    t->solve_assemble_waves = 2;
    t->hybrid_active_list = 1;
    t->far_batch_columns = 1;
}
// The caller's struct may be shorter than the library's (built against an older header): only its own bytes are written, and
// struct_size says how many those are.  A struct the library does not know (longer than its own) gets the bytes it does know.
void vf_engine_default_opts_sized(vf_engine_opts* o, uint32_t struct_size) {
    if (!o || struct_size < sizeof(uint32_t)) return;
    vf_engine_opts full;
    engine_defaults(&full);
    const uint32_t n = struct_size < sizeof(full) ? struct_size : (uint32_t)sizeof(full);
    memcpy(o, &full, n);
    o->struct_size = n;
}
void vf_engine_default_opts(vf_engine_opts* o) { vf_engine_default_opts_sized(o, (uint32_t)sizeof(vf_engine_opts)); }
void vf_engine_default_tuning_sized(vf_engine_tuning* t, uint32_t struct_size) {
    if (!t || struct_size < sizeof(uint32_t)) return;
    vf_engine_tuning full;
    tuning_defaults(&full);
    const uint32_t n = struct_size < sizeof(full) ? struct_size : (uint32_t)sizeof(full);
    memcpy(t, &full, n);
    t->struct_size = n;
}
void vf_engine_default_tuning(vf_engine_tuning* t) { vf_engine_default_tuning_sized(t, (uint32_t)sizeof(vf_engine_tuning)); }

int vf_engine_create(const vf_engine_opts* o, vf_engine** out) { return vf_engine_create_tuned(o, nullptr, out); }
// on_stream: the engine works on a stream somebody else owns (the column engine of a handle's far factors, the engine a growing
// handle moves into) -- hipStreamCreate costs 9 ms on this stack (the HSA queue behind it), more than everything else here together
static int create_engine(const vf_engine_opts* o_in, const vf_engine_tuning* t_in, bool on_stream, hipStream_t given, vf_engine** out);
int vf_engine_create_tuned(const vf_engine_opts* o_in, const vf_engine_tuning* t_in, vf_engine** out) {
    return create_engine(o_in, t_in, false, nullptr, out);
}
static int create_engine(const vf_engine_opts* o_in, const vf_engine_tuning* t_in, bool on_stream, hipStream_t given, vf_engine** out) {
    if (!o_in || !out) return fail(VF_ERR_INVALID, "null argument");
    // the structs as THIS library knows them: the caller's bytes over the defaults
    vf_engine_opts o_full;
    vf_engine_tuning t_full;
    engine_defaults(&o_full);
    tuning_defaults(&t_full);
    if (o_in->struct_size < 8 || o_in->struct_size > sizeof(o_full))
        return fail(VF_ERR_INVALID, "vf_engine_opts.struct_size = %u: this library knows sizes up to %zu (fill the struct with vf_engine_default_opts; "
                    "a caller built against a newer header needs a newer library)", o_in->struct_size, sizeof(o_full));
    memcpy(&o_full, o_in, o_in->struct_size);
    o_full.struct_size = (uint32_t)sizeof(o_full);
    if (t_in) {
        if (t_in->struct_size < 8 || t_in->struct_size > sizeof(t_full))
            return fail(VF_ERR_INVALID, "vf_engine_tuning.struct_size = %u: this library knows sizes up to %zu", t_in->struct_size, sizeof(t_full));
        memcpy(&t_full, t_in, t_in->struct_size);
        t_full.struct_size = (uint32_t)sizeof(t_full);
    }
    const vf_engine_opts* o = &o_full;
    const vf_engine_tuning* t = &t_full;
    if (o->windows < 1 || o->capacity < 2) return fail(VF_ERR_INVALID, "windows >= 1 and capacity >= 2 required");
    if (o->bandwidth < 1 || o->bandwidth > VF_MAX_BANDWIDTH)
        return fail(VF_ERR_INVALID, "bandwidth must be in 1..%d", VF_MAX_BANDWIDTH);
    if (o->chunks < 0 || o->chunks > 4096) return fail(VF_ERR_INVALID, "chunks must be in 0..4096");
    if (!(o->accept_rel >= 0.0) || !(o->accept_rel < 1.0)) return fail(VF_ERR_INVALID, "accept_rel must be in [0, 1)");
    if (o->refine_iterations < -1 || o->refine_iterations > 64) return fail(VF_ERR_INVALID, "refine_iterations must be in -1..64");
    if (!(o->gauge_floor >= 0.0) || !(o->gauge_floor < 1e6)) return fail(VF_ERR_INVALID, "gauge_floor must be in [0, 1e6)");
    if (o->lm_excursion < -1 || o->lm_excursion > 16) return fail(VF_ERR_INVALID, "lm_excursion must be in -1..16");
    if (!(o->refine_rel_stop >= 0.0) || !(o->refine_rel_stop < 1.0)) return fail(VF_ERR_INVALID, "refine_rel_stop must be in [0, 1)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(VF_ERR_NO_DEVICE, "no HIP device visible; libvilfusion has no CPU path");
    if (o->device < 0 || o->device >= ndev) return fail(VF_ERR_INVALID, "device %d out of range (%d)", o->device, ndev);
    HIPCHK(hipSetDevice(o->device));
    vf_engine* e = new vf_engine();
    e->opts = *o;
    e->tune = *t;
    vf::View& v = e->v;
    v.B = o->windows;
    v.M = (o->capacity + 63) / 64 * 64;
    v.G = (long)v.B * v.M;
    for (int i = 0; i < 3; i++) v.grav[i] = o->gravity[i];
    v.lam_up = o->lambda_up; v.lam_down = o->lambda_down; v.lam_min = o->lambda_min; v.lam_max = o->lambda_max;
    v.accept_rel = o->accept_rel;
    v.split_min = t->solve_split_min > 0 ? t->solve_split_min : 0;
    v.asm_min = t->solve_assemble_min > 0 ? t->solve_assemble_min : 0;
    v.asm_waves = t->solve_assemble_waves == 1 ? 1 : 2;
    if (on_stream) { e->stream = given; e->own_stream = false; }
    else HIPCHK(hipStreamCreate(&e->stream));
    HIPCHK(hipEventCreate(&e->ev0));
    HIPCHK(hipEventCreate(&e->ev1));
    const size_t G = (size_t)v.G, tiles = G / 64;
    int rc = VF_OK;
    {
        // everything the AL() lines below ask for, plus slack for their 256-byte alignment (alloc() falls back to allocations of
        // its own if this estimate is ever short)
        const size_t P0 = o->chunks >= 2 ? (size_t)o->chunks : (o->chunks == 0 && o->windows <= 128 ? 96 : 0);
        size_t dbl = G * (32 + vf::HROW + 15 + 15 + vf::PANEL) + tiles * 64 * (vf::IMU_IN + 2 * vf::IMU_R + vf::BTW_IN + 2 * vf::BTW_OUT) +
                     2 * (G / vf::JT) * vf::JT_STRIDE + 64 + (size_t)v.B * (vf::PRIOR_IN + 2 * vf::PRIOR_OUT + 48 + 729 + 27 + 56 + 8) + vf::HROW + vf::PLACE_CELLS;
        if (P0 >= 2) dbl += G * vf::VROW + (size_t)v.B * P0 * (vf::SEPK + vf::SEPL);
        if (o->incremental) dbl += (G >> vf::CK_LOG) * vf::CK_SZ;
        const size_t ints = G + 16 * (size_t)v.B + 64;
        if ((rc = e->reserve(dbl * sizeof(double) + ints * sizeof(int) + 64 * 256)) != VF_OK) { vf_engine_destroy(e); return rc; }
    }
#define AL(p, n) if ((rc = e->alloc(&(p), (n))) != VF_OK) { vf_engine_destroy(e); return rc; }
    AL(v.x, 2 * 16 * G);
    AL(v.imu_in, tiles * vf::IMU_IN * 64);
    AL(v.imu_r, 2 * tiles * vf::IMU_R * 64);
    AL(v.imu_j, 2 * (G / vf::JT) * vf::JT_STRIDE);
    AL(v.btw_a, G);
    AL(v.btw_in, tiles * vf::BTW_IN * 64);
    AL(v.btw_out, 2 * tiles * vf::BTW_OUT * 64);
    AL(v.prior_k, (size_t)v.B);
    AL(v.prior_in, (size_t)v.B * vf::PRIOR_IN);
    AL(v.prior_out, 2 * (size_t)v.B * vf::PRIOR_OUT);
    AL(v.mp_on, (size_t)v.B);
    AL(v.mp_x, (size_t)v.B * 48);
    AL(v.mp_L, (size_t)v.B * 729);
    AL(v.mp_eta, (size_t)v.B * 27);
    AL(v.mp_out, 2 * (size_t)v.B * 28);
    // K4 form: chunks = 0 picks it from the batch size: up to 128 windows -> partitioned solve with at most 96 chunks,
    // fewer on short windows (latency form); more windows -> one sweep per window (throughput form; the partitioned
    // solve does about twice the arithmetic).  1 forces sweeps.
    v.P = o->chunks >= 2 ? o->chunks : (o->chunks == 0 && o->windows <= 128 ? 96 : 0);
    v.P_fit = o->chunks == 0 ? 1 : 0;
    // (the chunk kernels launch windows x P workgroups: no more chunks than a full window could use)
    if (v.P_fit && v.P) v.P = vf::chunk_count(v.M, v.P, 1);
    if (v.P < 2) v.P = 0;
    AL(v.H, G * vf::HROW);
    AL(v.gvec, G * 15 + 64); // + slack: the solver's row fetch reads 64 lanes of a 15-double row (the excess is never used)
    AL(v.place, vf::PLACE_CELLS);
    AL(v.zrow, vf::HROW);         // a block row of zeros: what the solver fetches for rows outside the window
    AL(v.delta, G * 15 + (size_t)v.B);   // + one solve-failure flag per window (time-sharded windows: reduced with the increments)
    AL(v.Lp, G * vf::PANEL);
    if (v.P) {
        const size_t BP = (size_t)v.B * v.P;
        AL(v.Vp, G * vf::VROW);
        AL(v.sepR, BP * vf::SEPK);       // one buffer, slots of [sepR | sepS | sepC] (vf_kernels.hpp "SEPK")
        v.sepS = v.sepR + vf::SEPM;
        v.sepC = v.sepR + 2 * vf::SEPM;
        AL(v.sepL, BP * vf::SEPL);
    }
    AL(v.lo, (size_t)v.B);
    AL(v.hi, (size_t)v.B);
    AL(v.sel, (size_t)v.B);
    AL(v.fail, (size_t)v.B);
    AL(v.fresh, (size_t)v.B);
    AL(v.lambda, (size_t)v.B);
    AL(v.cost, (size_t)v.B);
    AL(v.n_acc, (size_t)v.B);
    AL(v.n_rej, (size_t)v.B);
    AL(v.n_fail, (size_t)v.B);
    AL(v.done, (size_t)v.B);
    AL(v.n_active, 4);
    v.gate = 0;
    v.gate_T = t->hybrid_threshold;
    v.tw_max = t->sweep_two_sided_max;
    v.stop_on = 0;
    v.rel_tol = v.abs_tol = 0.0;
    v.sh_r = 0;
    v.sh_G = 1;
    v.gauge_floor = o->gauge_floor;
    v.sh_all_jac = 0;         // set at every linearisation: 1 while the engine refines (the refined solve applies J on whole increments)
    v.wildfire = o->wildfire >= 0.0 ? o->wildfire : 0.0;
    if (o->incremental) {
        AL(v.inc_k, (size_t)v.B);
        AL(v.inc_stop, (size_t)v.B);
        AL(v.inc_from, (size_t)v.B);
        AL(v.ck, (G >> vf::CK_LOG) * vf::CK_SZ);
        HIPCHK(hipMemsetAsync(v.inc_k, 0x7f, v.B * sizeof(int), e->stream));
    }
    AL(e->lambda0_dev, (size_t)v.B);
    AL(e->sigma_dev, 16);
    AL(e->status_dev, 4);
#undef AL
    HIPCHK(hipMemsetAsync(v.btw_a, 0xff, G * sizeof(int), e->stream));   // -1 = empty slot
    HIPCHK(hipMemsetAsync(v.prior_k, 0xff, v.B * sizeof(int), e->stream));
    std::vector<double> lam((size_t)v.B, o->lambda0);
    HIPCHK(hipMemcpyAsync(v.lambda, lam.data(), v.B * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->lambda0_dev, lam.data(), v.B * sizeof(double), hipMemcpyHostToDevice, e->stream));
    // measured on MI355X (ROCm 7.0, one 1000-pose window, K = 5): 2.99 ms replayed from the graph vs 2.91 ms
    // with plain asynchronous launches -- the queue is never empty, so there is no launch gap to remove.
    // Hence opt-in only.
    e->graph_off = t->use_hip_graph == 0;
    e->no_warm = o->cold_start != 0;
    HIPCHK(hipStreamSynchronize(e->stream));
    e->h_lo.assign(v.B, 0);
    e->h_hi.assign(v.B, 0);
    *out = e;
    return VF_OK;
}

void vf_engine_destroy(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (!e) return;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    e->drop_graph();
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->stage) (void)hipFree(e->stage);
    if (e->pre_buf) (void)hipFree(e->pre_buf);
    if (e->in_host) (void)hipHostFree(e->in_host);
    if (e->rq_stop_host) (void)hipHostFree(e->rq_stop_host);
    if (e->in_dev) (void)hipFree(e->in_dev);
    if (e->in_status) (void)hipFree(e->in_status);
    for (auto ev : e->in_ev) if (ev) (void)hipEventDestroy(ev);
    if (e->far_columns) { vf_engine_destroy(e->far_columns); e->far_columns = nullptr; }
    if (e->stream2) { (void)hipStreamSynchronize(e->stream2); (void)hipStreamDestroy(e->stream2); }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->sticky_dev) (void)hipFree(e->sticky_dev);
    if (e->res_host) (void)hipHostFree(e->res_host);
    if (e->x_gtmp) (void)hipFree(e->x_gtmp);
    if (e->x_Z) (void)hipFree(e->x_Z);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->stream && e->own_stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

// Warm start across the GraphManager's call sequence (vf_solve: preintegrate / predict / set_between for the NEW keyframes,
// marginalize + drop_oldest for the ones that leave, set_range): a call that only touches keyframe slots at or beyond the
// window's current end leaves every existing record, H row and g entry those of the current states, exactly as
// vf_engine_slide does -- the next vf_engine_iterate then linearises only what was appended (k_linearize_tail).  Engines
// with one window only (`slid` is one count for the whole engine); anything else makes the next solve a cold start.
// A factor record written a few slots INSIDE the window's end (late odometry for a keyframe an earlier solve already
// covered) lengthens the tail that is linearised again (`redo`), up to the 8 keyframes k_linearize_tail handles.
static void touch(vf_engine* e, int window, int first_slot) {
    if (!e) return;
    if (e->inc_valid && e->v.B == 1 && window == 0 && first_slot > e->h_lo[0]) {
        // (incremental engines follow a write anywhere behind the window's first keyframe: the update starts in front of it)
        if (first_slot < e->inc_first_dirty) e->inc_first_dirty = first_slot;
        if (!e->warm) return;
        const int inside = e->h_hi[0] - first_slot;
        if (inside > e->redo) e->redo = inside;
        if (e->redo > 8) e->warm = false;
        return;
    }
    if (e->warm && e->v.B == 1 && window == 0 && first_slot > e->h_lo[0]) {
        const int inside = e->h_hi[0] - first_slot;          // <= 0: beyond the current end
        if (inside > e->redo) e->redo = inside;
        if (e->redo <= 8) return;
    }
    cold(e);
}
static int not_sharded_(vf_engine* e, const char* what) {
    if (e && e->v.sh_G > 1) return fail(VF_ERR_INVALID, "%s: not for time-sharded engines", what);
    return VF_OK;
}
static int check_window(vf_engine* e, int window) {
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (window < 0 || window >= e->v.B) return fail(VF_ERR_INVALID, "window %d out of range", window);
    return VF_OK;
}
static int check_range(vf_engine* e, int window, int k0, int n) {
    int rc = check_window(e, window);
    if (rc) return rc;
    if (k0 < 0 || n < 0 || k0 + n > e->v.M) return fail(VF_ERR_BAD_KEY, "keyframes [%d,%d) outside capacity %d", k0, k0 + n, e->v.M);
    return VF_OK;
}

int vf_engine_set_range(vf_engine* e, int window, int lo, int hi) {
    DeviceGuard dev_guard_(e);
    int rc = check_window(e, window);
    if (rc) { if (e) cold(e); return rc; }
    if (lo < 0 || hi < lo || hi > e->v.M) { cold(e); return fail(VF_ERR_BAD_KEY, "bad range [%d,%d)", lo, hi); }
    // growing the end of the one window of a warm engine = appending keyframes (see touch()); its start is moved by
    // vf_engine_drop_oldest only
    const bool grows = e->v.B == 1 && lo == e->h_lo[0] && hi >= e->h_hi[0] && e->h_hi[0] > e->h_lo[0];
    const bool inc_keeps = grows && e->inc_valid;
    if (e->warm && grows) e->slid += hi - e->h_hi[0];
    else cold(e);
    if (inc_keeps) {
        e->inc_valid = true;
        if (hi > e->h_hi[0] && e->h_hi[0] < e->inc_first_dirty) e->inc_first_dirty = e->h_hi[0];
    }
    if (e->async_now()) {
        vf::launch_set_range(e->v, window, lo, hi, e->stream);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpyAsync(e->v.lo + window, &lo, sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->v.hi + window, &hi, sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    e->h_lo[window] = lo;
    e->h_hi[window] = hi;
    return VF_OK;
}
int vf_engine_set_async(vf_engine* e, int on) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (on) { if (int rc = e->ensure_async()) return rc; }
    e->async_on = on != 0;
    return VF_OK;
}
int vf_engine_read_result(vf_engine* e, int window, int slot, int estimate, double* state16, double* cost, int* accepted, int* rejected,
                          int* solve_failures, int* device_flags) {
    DeviceGuard dev_guard_(e);
    int rc = check_range(e, window, slot, 1);
    if (rc) return rc;
    if ((rc = e->ensure_async())) return rc;
    vf::launch_read_result(e->v, window, slot, estimate ? 1 : 0, e->sticky_dev, e->res_host, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    const vf::SolveResult& r = *e->res_host;
    if (state16) memcpy(state16, r.state, sizeof(r.state));
    if (cost) *cost = r.cost;
    if (accepted) *accepted = r.n_acc;
    if (rejected) *rejected = r.n_rej;
    if (solve_failures) *solve_failures = r.n_fail;
    if (device_flags) *device_flags = (r.sticky[0] ? 1 : 0) | (r.sticky[1] & 4 ? 4 : (r.sticky[1] ? 2 : 0));
    return VF_OK;
}

int vf_engine_set_states(vf_engine* e, int window, int k0, int n, const double* s) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (!s) return fail(VF_ERR_INVALID, "null states");
    if (n == 0) return VF_OK;
    const size_t bytes = (size_t)n * 16 * sizeof(double);
    if ((rc = e->ensure_stage(bytes))) return rc;
    HIPCHK(hipMemcpyAsync(e->stage, s, bytes, hipMemcpyHostToDevice, e->stream));
    const long g0 = (long)window * e->v.M + k0;
    // both buffers get the value so that "current" is well defined whatever sel is
    vf::launch_scatter_states(e->stage, e->v.x, e->v.G, 0, g0, n, e->stream);
    vf::launch_scatter_states(e->stage, e->v.x, e->v.G, 1, g0, n, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

int vf_engine_get_states(vf_engine* e, int window, int k0, int n, double* s) {
    DeviceGuard dev_guard_(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (!s) return fail(VF_ERR_INVALID, "null states");
    if (n == 0) return VF_OK;
    const size_t bytes = (size_t)n * 16 * sizeof(double);
    if ((rc = e->ensure_stage(bytes))) return rc;
    const long g0 = (long)window * e->v.M + k0;
    vf::launch_gather_states(e->v.x, e->stage, e->v.G, e->v.sel, e->v.M, 0, g0, n, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s, e->stage, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

int vf_engine_set_imu(vf_engine* e, int window, int k0, int n, const double* rec) {
    DeviceGuard dev_guard_(e);
    touch(e, window, k0);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (!rec) return fail(VF_ERR_INVALID, "null records");
    if (n == 0) return VF_OK;
    const size_t bytes = (size_t)n * vf::IMU_IN * sizeof(double);
    if ((rc = e->ensure_stage(bytes))) return rc;
    HIPCHK(hipMemcpyAsync(e->stage, rec, bytes, hipMemcpyHostToDevice, e->stream));
    vf::launch_scatter(e->stage, e->v.imu_in, (long)window * e->v.M + k0, n, vf::IMU_IN, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

int vf_engine_set_between(vf_engine* e, int window, int n, const int32_t* a, const int32_t* b, const double* rec) {
    DeviceGuard dev_guard_(e, true);
    {
        int first = 1 << 30;
        for (int i = 0; i < n && b; i++) first = b[i] < first ? b[i] : first;
        touch(e, window, n > 0 ? first : 0);
    }
    int rc = check_window(e, window);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!a || !b || !rec))) return fail(VF_ERR_INVALID, "null argument");
    const int M = e->v.M, W = e->opts.bandwidth;
    for (int i = 0; i < n; i++) {
        if (a[i] < 0 || b[i] >= M || a[i] >= b[i]) return fail(VF_ERR_BAD_KEY, "between factor %d: need 0 <= a < b < capacity (a=%d b=%d)", i, a[i], b[i]);
        if (b[i] - a[i] > W) return fail(VF_ERR_CAPACITY, "between factor %d spans %d keyframes > bandwidth %d", i, b[i] - a[i], W);
    }
    if (e->async_now() && n <= 4) {
        // (a record travels as a kernel argument: no staging copy, no synchronisation)
        for (int i = 0; i < n; i++) {
            vf::BtwArg arg;
            memcpy(arg.r, rec + (size_t)i * vf::BTW_IN, sizeof(arg.r));
            vf::launch_put_between(e->v, (long)window * M + b[i], a[i], arg, e->stream);
        }
        HIPCHK(hipGetLastError());
        return VF_OK;
    }
    if (e->side_open) { if (int rj = e->join_side()) return rj; }
    // records go to the slot of b: runs of consecutive b are staged in one scatter
    int i = 0;
    while (i < n) {
        int j = i + 1;
        while (j < n && b[j] == b[j - 1] + 1) j++;
        const int cnt = j - i;
        const size_t bytes = (size_t)cnt * vf::BTW_IN * sizeof(double);
        if ((rc = e->ensure_stage(bytes))) return rc;
        HIPCHK(hipMemcpyAsync(e->stage, rec + (size_t)i * vf::BTW_IN, bytes, hipMemcpyHostToDevice, e->stream));
        const long g0 = (long)window * M + b[i];
        vf::launch_scatter(e->stage, e->v.btw_in, g0, cnt, vf::BTW_IN, e->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(e->v.btw_a + g0, a + i, cnt * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        i = j;
    }
    return VF_OK;
}

int vf_engine_set_extra_between(vf_engine* e, int window, int n, const int32_t* a, const int32_t* b, const double* rec) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    int rc = check_window(e, window);
    if (rc) return rc;
    if (int rs = not_sharded_(e, "vf_engine_set_extra_between")) return rs;
    if (n < 0 || n + e->h_ln(window) > VF_MAX_EXTRA)
        return fail(VF_ERR_CAPACITY, "at most %d far between factors per window (got %d, and %d carried on from keyframes that have been marginalised)",
                    VF_MAX_EXTRA, n, e->h_ln(window));
    if (n > 0 && (!a || !b || !rec)) return fail(VF_ERR_INVALID, "null argument");
    const int M = e->v.M, B = e->v.B, X = VF_MAX_EXTRA;
    for (int i = 0; i < n; i++) {
        if (a[i] < 0 || b[i] >= M || a[i] >= b[i]) return fail(VF_ERR_BAD_KEY, "far between factor %d: need 0 <= a < b < capacity (a=%d b=%d)", i, a[i], b[i]);
        if (!(rec[(size_t)i * vf::BTW_IN + 7] > 0.0)) return fail(VF_ERR_NOT_SPD, "far between factor %d: singular square-root information", i);
    }
    if (e->v.x_max == 0 && n == 0) return VF_OK;
    static const bool timing = getenv("VF_SOLVE_TIMING") != nullptr;
    auto t_a = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (timing) { auto t_b = std::chrono::steady_clock::now(); fprintf(stderr, "[set_extra] %-14s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t_b - t_a).count()); t_a = t_b; } };
    {
        int need = n + e->h_ln(window);                   // slots in use once this call is through
        for (int w = 0; w < B && e->v.x_max; w++) if (w != window) need = std::max(need, e->h_xn[w] + e->h_ln(w));
        if ((rc = e->ensure_far(need))) return rc;
    }
    e->attach_far();
    lap("ensure_far");
    // the column engine holds a copy of the window's H per Woodbury column: sized by the far factors alive (6 columns each), grown
    // by doubling -- one loop closure costs 6 windows (0.14 GB, 2 ms to make at 1 200 slots), not the 48 of VF_MAX_EXTRA (1.3 GB, 13 ms)
    const int far_need = n + e->h_ln(window);
    if (e->far_columns && e->far_columns->v.B < 6 * far_need) { vf_engine_destroy(e->far_columns); e->far_columns = nullptr; }
    if (n > 0 && !e->far_columns && !e->is_far_columns && B == 1 && e->v.P >= 2 && e->tune.far_batch_columns) {
        vf_engine_opts co = e->opts;
        int slots = 1;
        while (slots < far_need) slots *= 2;
        co.windows = 6 * (slots < VF_MAX_EXTRA ? slots : VF_MAX_EXTRA);
        co.capacity = M;
        co.incremental = 0;
        vf_engine_tuning ct = e->tune;
        ct.use_hip_graph = 0;
        vf_engine* c = nullptr;
        if (create_engine(&co, &ct, true, e->stream, &c) == VF_OK) {
            lap("create columns");
            c->is_far_columns = true;
            if (c->v.P == e->v.P && c->v.P_fit == e->v.P_fit) e->far_columns = c;
            else vf_engine_destroy(c);
        }
    }
    e->h_xa[window].assign(a, a + n);
    e->h_xb[window].assign(b, b + n);
    e->h_xrec[window].assign(rec, rec + (size_t)n * vf::BTW_IN);
    std::vector<int> ha(X, -1), hb(X, -1);
    std::vector<double> hr((size_t)X * vf::BTW_IN, 0.0);
    for (int i = 0; i < n; i++) { ha[i] = a[i]; hb[i] = b[i]; }
    if (n) memcpy(hr.data(), rec, (size_t)n * vf::BTW_IN * sizeof(double));
    HIPCHK(hipMemcpyAsync(e->v.x_a + (size_t)window * X, ha.data(), X * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->v.x_b + (size_t)window * X, hb.data(), X * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->v.x_in + (size_t)window * X * vf::BTW_IN, hr.data(), hr.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
    // both linearisation buffers of the window start from zeros (an empty slot must read as "no factor")
    for (int bf = 0; bf < 2; bf++)
        HIPCHK(hipMemsetAsync(e->v.x_out + ((size_t)bf * B + window) * X * vf::BTW_OUT, 0, (size_t)X * vf::BTW_OUT * sizeof(double), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    lap("lists");
    e->h_xn[window] = n;
    e->recount_far();
    e->epoch++;        // (the number of band solves per trial is baked into a captured launch sequence)
    return VF_OK;
}

int vf_engine_clear_between(vf_engine* e, int window, int k0, int n) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (n == 0) return VF_OK;
    HIPCHK(hipMemsetAsync(e->v.btw_a + (long)window * e->v.M + k0, 0xff, n * sizeof(int), e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

int vf_engine_set_prior(vf_engine* e, int window, int k, const double* rec) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    int rc = check_range(e, window, k, 1);
    if (rc) return rc;
    if (!rec) return fail(VF_ERR_INVALID, "null record");
    for (int i = 0; i < 15; i++)
        if (!(rec[16 + i] > 0.0)) return fail(VF_ERR_NOT_SPD, "prior sigma %d must be > 0", i);
    HIPCHK(hipMemcpyAsync(e->v.prior_in + (size_t)window * vf::PRIOR_IN, rec, vf::PRIOR_IN * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->v.prior_k + window, &k, sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(e->v.mp_on + window, 0, sizeof(int), e->stream));   // a fresh anchor replaces any marginal prior
    if (e->v.x_max > 0 && e->h_ln(window) > 0) {                               // ... and the linear far factor that is expressed around it
        HIPCHK(hipMemsetAsync(e->v.xl_n + window, 0, sizeof(int), e->stream));
        e->far_ended += e->h_ln(window);
        e->h_lb[window].clear();
        e->recount_far();
        e->epoch++;
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

int vf_engine_preintegrate(vf_engine* e, int window, int k0, int n, const int32_t* off, const double* steps,
                           const double* bhat, const vf_imu_params* p) {
    DeviceGuard dev_guard_(e, true);
    touch(e, window, k0);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (!off || !bhat || !p) return fail(VF_ERR_INVALID, "null argument");
    if (n == 0) return VF_OK;
    if (k0 < 1) return fail(VF_ERR_BAD_KEY, "imu factor slot 0 does not exist (factor k links k-1 -> k)");
    const int total = off[n];
    if (off[0] < 0 || total < off[0] || (total > 0 && !steps)) return fail(VF_ERR_INVALID, "bad step offsets");
    for (int i = 0; i < n; i++)
        if (off[i + 1] <= off[i]) return fail(VF_ERR_INDETERMINATE, "imu factor for keyframe %d has no IMU steps", k0 + i);
    if (e->async_now() && k0 >= e->h_hi[window]) {
        // asynchronous staging: the arguments travel through the pinned block of vf_engine_ingest_tail in ONE copy, the status
        // goes into the sticky word (vf_engine_read_result); nothing here waits for the device
        const size_t steps_b = (size_t)(total > 0 ? total : 1) * 7 * sizeof(double), bias_b = (size_t)n * 6 * sizeof(double);
        const size_t off_b = ((size_t)(n + 1) * sizeof(int) + 7) & ~(size_t)7, bytes = steps_b + bias_b + off_b;
        if ((rc = e->ensure_ingest(bytes))) return rc;
        if (e->in_pending) HIPCHK(hipEventSynchronize(e->in_ev[1]));      // (the previous copy has left the pinned buffer: long done)
        char* h = (char*)e->in_host;
        if (total > 0) memcpy(h, steps, (size_t)total * 7 * sizeof(double));
        memcpy(h + steps_b, bhat, bias_b);
        memcpy(h + steps_b + bias_b, off, (size_t)(n + 1) * sizeof(int));
        char* d = (char*)e->in_dev;
        HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipEventRecord(e->in_ev[1], e->stream));
        e->in_pending = true;
        vf::ImuCov c{p->acc_cov, p->gyro_cov, p->integration_cov, p->bias_acc_cov, p->bias_omega_cov, p->bias_acc_omega_int};
        vf::launch_preintegrate(e->v, (long)window * e->v.M + k0, n, (const int*)(d + steps_b + bias_b), (const double*)d, (const double*)(d + steps_b), c,
                                e->sticky_dev, e->stream);
        HIPCHK(hipGetLastError());
        return VF_OK;
    }
    if (e->side_open) { if (int rj = e->join_side()) return rj; }
    // one scratch block: [steps 7 x total][bias 6 x n][offsets n + 1][status]
    const size_t steps_b = (size_t)(total > 0 ? total : 1) * 7 * sizeof(double), bias_b = (size_t)n * 6 * sizeof(double);
    const size_t off_b = ((size_t)(n + 1) * sizeof(int) + 7) & ~(size_t)7;
    if ((rc = e->ensure_pre(steps_b + bias_b + off_b + 8))) return rc;
    double* d_steps = (double*)e->pre_buf;
    double* d_bhat = (double*)((char*)e->pre_buf + steps_b);
    int* d_off = (int*)((char*)e->pre_buf + steps_b + bias_b);
    int* d_status = (int*)((char*)e->pre_buf + steps_b + bias_b + off_b);
    HIPCHK(hipMemcpyAsync(d_off, off, (n + 1) * sizeof(int), hipMemcpyHostToDevice, e->stream));
    if (total > 0) HIPCHK(hipMemcpyAsync(d_steps, steps, (size_t)total * 7 * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(d_bhat, bhat, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemsetAsync(d_status, 0, sizeof(int), e->stream));
    vf::ImuCov c{p->acc_cov, p->gyro_cov, p->integration_cov, p->bias_acc_cov, p->bias_omega_cov, p->bias_acc_omega_int};
    vf::launch_preintegrate(e->v, (long)window * e->v.M + k0, n, d_off, d_steps, d_bhat, c, d_status, e->stream);
    HIPCHK(hipGetLastError());
    int status = 0;
    HIPCHK(hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (status) return fail(VF_ERR_NOT_SPD, "preintegrated covariance not positive definite");
    return VF_OK;
}

int vf_engine_ingest_tail(vf_engine* e, const int32_t* step_off, const double* steps, const vf_imu_params* p,
                          const int32_t* btw_a, const double* btw_rec) {
    DeviceGuard dev_guard_(e);
    if (!e || !step_off || !p || !btw_a || !btw_rec) return fail(VF_ERR_INVALID, "null argument");
    const int B = e->v.B, M = e->v.M, W = e->opts.bandwidth;
    const int total = step_off[B];
    if (step_off[0] != 0 || (total > 0 && !steps)) return fail(VF_ERR_INVALID, "bad step offsets");
    for (int w = 0; w < B; w++) {
        if (step_off[w + 1] <= step_off[w]) return fail(VF_ERR_INDETERMINATE, "window %d: the new IMU factor has no IMU steps", w);
        const int hi = e->h_hi[w], lo = e->h_lo[w];
        if (hi >= M) return fail(VF_ERR_CAPACITY, "window %d has no free keyframe slot", w);
        if (hi <= lo) return fail(VF_ERR_BAD_KEY, "window %d is empty", w);
        if (btw_a[w] >= 0 && (btw_a[w] >= hi || btw_a[w] < lo)) return fail(VF_ERR_BAD_KEY, "window %d: between factor from keyframe %d outside the window [%d, %d)", w, btw_a[w], lo, hi);
        if (btw_a[w] >= 0 && hi - btw_a[w] > W) return fail(VF_ERR_CAPACITY, "window %d: between factor spans %d keyframes > bandwidth %d", w, hi - btw_a[w], W);
    }
    // one packed block: [offsets B + 1][between sources B][between records 28 B][steps 7 x total]
    const size_t off_b = ((size_t)(B + 1) * sizeof(int) + 7) & ~(size_t)7, a_b = ((size_t)B * sizeof(int) + 7) & ~(size_t)7;
    const size_t rec_b = (size_t)B * vf::BTW_IN * sizeof(double), st_b = (size_t)total * 7 * sizeof(double);
    const size_t bytes = off_b + a_b + rec_b + st_b;
    int rc = e->ensure_ingest(bytes);
    if (rc) return rc;
    if (e->in_pending) HIPCHK(hipEventSynchronize(e->in_ev[1]));      // the previous call's copy has left the pinned buffer
    char* h = (char*)e->in_host;
    memcpy(h, step_off, (size_t)(B + 1) * sizeof(int));
    memcpy(h + off_b, btw_a, (size_t)B * sizeof(int));
    memcpy(h + off_b + a_b, btw_rec, rec_b);
    memcpy(h + off_b + a_b + rec_b, steps, st_b);
    char* d = (char*)e->in_dev;
    HIPCHK(hipEventRecord(e->in_ev[0], e->stream));
    HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipEventRecord(e->in_ev[1], e->stream));
    vf::ImuCov c{p->acc_cov, p->gyro_cov, p->integration_cov, p->bias_acc_cov, p->bias_omega_cov, p->bias_acc_omega_int};
    vf::launch_ingest_tail(e->v, (const int*)d, (const double*)(d + off_b + a_b + rec_b), (const int*)(d + off_b),
                           (const double*)(d + off_b + a_b), c, e->in_status, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e->in_ev[2], e->stream));
    e->in_pending = true;
    // Nothing inside any window changed (the records sit in slot hi, beyond every window's end): a warm engine stays warm,
    // the vf_engine_slide that follows appends the keyframe and the next solve linearises it (k_linearize_tail).
    return VF_OK;
}

int vf_engine_ingest_status(vf_engine* e, float* h2d_ms, float* k0_ms) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (h2d_ms) *h2d_ms = 0.f;
    if (k0_ms) *k0_ms = 0.f;
    if (!e->in_pending) return VF_OK;
    HIPCHK(hipEventSynchronize(e->in_ev[2]));
    if (h2d_ms) HIPCHK(hipEventElapsedTime(h2d_ms, e->in_ev[0], e->in_ev[1]));
    if (k0_ms) HIPCHK(hipEventElapsedTime(k0_ms, e->in_ev[1], e->in_ev[2]));
    int status = 0;
    HIPCHK(hipMemcpy(&status, e->in_status, sizeof(int), hipMemcpyDeviceToHost));
    if (status) {
        HIPCHK(hipMemset(e->in_status, 0, sizeof(int)));
        if (status & 2) return fail(VF_ERR_CAPACITY, "vf_engine_ingest_tail: a window had no free keyframe slot");
        return fail(VF_ERR_NOT_SPD, "vf_engine_ingest_tail: a preintegrated covariance was not positive definite");
    }
    return VF_OK;
}

int vf_engine_get_imu(vf_engine* e, int window, int k0, int n, double* rec) {
    DeviceGuard dev_guard_(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (n == 0 || !rec) return VF_OK;
    const size_t bytes = (size_t)n * vf::IMU_IN * sizeof(double);
    if ((rc = e->ensure_stage(bytes))) return rc;
    vf::launch_gather(e->v.imu_in, e->stage, (long)window * e->v.M + k0, n, vf::IMU_IN, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(rec, e->stage, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

// ------------------------------------------------------------------ stages
int vf_engine_linearize(vf_engine* e, int which) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    which = which ? 1 : 0;
    // time-sharded ranks write the Jacobian of every factor only while the solve is refined (windows longer than
    // refine_min_keyframes, or refine_iterations > 0): otherwise each writes what feeds its own rows, and K1's traffic shrinks
    // with the world size
    e->v.sh_all_jac = e->refine_iters() > 0 ? 1 : 0;
    // a new linearisation of the CURRENT states invalidates H, g and starts a new solve (no window is converged yet)
    if (!which) {
        HIPCHK(hipMemsetAsync(e->v.fresh, 0x01, e->v.B * sizeof(int), e->stream));
        HIPCHK(hipMemsetAsync(e->v.done, 0, e->v.B * sizeof(int), e->stream));
    }
    if (e->v.B <= 128 || e->v.sh_G > 1) {
        vf::launch_linearize_all(e->v, which, e->stream);       // latency form: K1, K2, K2b side by side
    } else {
        vf::launch_linearize_imu(e->v, which, e->stream);
        vf::launch_linearize_between_prior(e->v, which, e->stream);
    }
    if (e->x_used > 0) vf::launch_linearize_extra(e->v, which, e->stream);
    HIPCHK(hipGetLastError());
    return VF_OK;
}
// the next vf_engine_solve forms the normal equations inside its forward sweep (vf_engine_opts.solve_assemble_min): K3 has
// nothing to do.  Not while far factors are held (their correction solves from H and g) nor in the hybrid form.
static bool assembles_in_solve(const vf_engine* e) {
    return vf::asm_in_solve(e->v) && e->x_used == 0 && !(e->hybrid && e->v.stop_on) && e->refine_iters() == 0;
}
// hybrid solves (termination rule on): the sweep half assembles its own rows, K3 runs for the partitioned half only
static bool assembles_in_hybrid(const vf_engine* e) {
    return e->hybrid && e->v.stop_on && vf::asm_in_hybrid(e->v) && e->x_used == 0 && e->refine_iters() == 0;
}
int vf_engine_assemble(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (assembles_in_solve(e)) return VF_OK;
    if (assembles_in_hybrid(e)) {
        vf::launch_assemble_for_partitioned(e->v, e->stream);
        HIPCHK(hipGetLastError());
        return VF_OK;
    }
    vf::launch_assemble(e->v, e->stream);
    if (e->x_used > 0) vf::launch_extra_gradient(e->v, e->stream);   // the far factors' J^T r (their J^T J stays out of the band)
    HIPCHK(hipGetLastError());
    return VF_OK;
}
static int not_sharded(vf_engine* e, const char* what) {
    if (e->v.sh_G > 1)
        return fail(VF_ERR_INVALID, "%s works on whole windows; this engine holds shard %d of %d (use the staged calls, "
                    "include/vilfusion.h \"time-sharded windows\")", what, e->v.sh_r, e->v.sh_G);
    return VF_OK;
}
int vf_engine_solve(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (int rc = not_sharded(e, "vf_engine_solve")) return rc;
    auto band_solve = [&](double* gvec, double* delta, const int* skip = nullptr) {      // the engine's K4 form on another right-hand side / increment buffer
        vf::View a = e->v;
        a.gvec = gvec;
        a.delta = delta;
        // (skip: windows that take no part -- a refinement's correction solves pass its stop flags, which include the windows
        // the termination rule has finished, so that a window whose corrections have converged costs its later solves nothing)
        if (skip) { a.stop_on = 1; a.done = const_cast<int*>(skip); }
        if (!assembles_in_solve(e) && !assembles_in_hybrid(e)) a.asm_min = 0;
        if (e->hybrid && e->v.stop_on) {
            vf::View p = e->partitioned_view();
            p.gvec = gvec;
            p.delta = delta;
            if (skip) { p.stop_on = 1; p.done = const_cast<int*>(skip); }     // (the partitioned half skips the same windows)
            a.act = e->act_list;
            vf::launch_band_solve_hybrid(a, p, e->stream);
        } else vf::launch_band_solve(a, e->stream);
    };
    band_solve(e->v.gvec, e->v.delta);
    const double* far_Z = e->x_Z;
    size_t far_zstride = e->x_zstride;
    if (e->x_used > 0) {
        // Far between factors: (H_band + lambda I + U U^T) delta = -g by Woodbury -- the band solver once more per column of
        // U (6 per slot in use; it refactorises every time: a fallback for the rare window with such factors, not a fast
        // path), then one small dense system per window (k_extra_combine).
        if (e->far_columns) {
            // (single-window engines: every column in ONE partitioned solve of the column engine -- same kernels, same bits)
            const vf::View& c = e->far_columns->v;
            vf::launch_cols_prepare(e->v, c, 6 * e->x_used, e->stream);
            vf::launch_partitioned_solve(c, e->stream);
            vf::launch_cols_fail(e->v, c, 6 * e->x_used, e->stream);
            far_Z = c.delta;
            far_zstride = (size_t)c.M * 15;
        } else {
            const size_t gbytes = ((size_t)e->v.G * 15 + 64) * sizeof(double);
            for (int s = 0; s < e->x_used; s++)
                for (int j = 0; j < 6; j++) {
                    HIPCHK(hipMemsetAsync(e->x_gtmp, 0, gbytes, e->stream));
                    vf::launch_extra_rhs(e->v, s, j, e->x_gtmp, e->stream);
                    band_solve(e->x_gtmp, e->x_Z + (size_t)(6 * s + j) * e->x_zstride);
                }
        }
        vf::launch_extra_combine(e->v, far_Z, far_zstride, e->x_used, e->stream);   // (slots beyond x_used are empty in every window)
    }
    if (const int R = e->refine_iters()) {
        // Refined solve (vf_refine.hip): the increment just computed is the start, the factorisation the preconditioner, of
        // conjugate gradients on the normal equations with the operator applied through J -- R correction solves.  With far
        // factors the operator has their rows too (k_far_apply) and the preconditioner is the Woodbury solve above: every
        // correction is one band solve combined with the columns Z that are already there.  (The band factor alone will not
        // do as the preconditioner: against the 1e-10 curvature of a long window's soft modes a loop closure's information is
        // an eigenvalue of 1e14 in M^-1 A, and the band-only start answers the closure's gradient with a step of kilometres.)
        if (int rc = e->ensure_refine()) return rc;
        vf::launch_refine_begin(e->v, e->rq, e->stream);
        for (int it = 0; it < R; it++) {
            band_solve(e->rq.nres, e->rq.z, e->rq.stop);
            if (e->x_used > 0) {
                vf::View a = e->v;
                a.delta = e->rq.z;
                a.stop_on = 1;
                a.done = e->rq.stop;
                vf::launch_extra_combine(a, far_Z, far_zstride, e->x_used, e->stream);
            }
            vf::launch_refine_step(e->v, e->rq, e->refine_stop(), e->stream);
            // How many corrections a window needs grows with its length (4 at 1 600 keyframes, 12 at 10 000): after the
            // 4th, 6th, ... the stop flags are read back, and once every window has stopped the rest are not issued
            // (a skipped correction is ~15 empty launches; the read-back costs one stream synchronisation)
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(e->stream, &capturing);      // (a caller capturing its own stream: no read-back, every correction is issued)
            if (it >= 1 && it % 2 == 1 && it + 1 < R && capturing == hipStreamCaptureStatusNone) {
                HIPCHK(hipMemcpyAsync(e->rq_stop_host, e->rq.stop, e->v.B * sizeof(int), hipMemcpyDeviceToHost, e->stream));
                HIPCHK(hipStreamSynchronize(e->stream));
                bool live = false;
                for (int w = 0; w < e->v.B && !live; w++) live = e->rq_stop_host[w] == 0;
                if (!live) break;
            }
        }
        vf::launch_refine_end(e->v, e->rq, e->stream);
    }
    HIPCHK(hipGetLastError());
    return VF_OK;
}
// ---- the refined solve on time-sharded engines, staged like the solve itself: after the trial's two collectives
//   vf_engine_refine_begin ; R x { vf_engine_solve_local, <all-gather sep>, vf_engine_solve_global, <all-reduce refine_delta>,
//   vf_engine_refine_step } ; vf_engine_refine_end
// (vf_engine_solve_local / _global work on the correction's right-hand side while a refinement is open)
int vf_engine_refine_count(vf_engine* e, int* iterations) {
    if (!e || !iterations) return fail(VF_ERR_INVALID, "null argument");
    *iterations = e->refine_iters();
    return VF_OK;
}
int vf_engine_refine_begin(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    cold(e);
    if (int rc = e->ensure_refine()) return rc;
    vf::launch_refine_begin(e->v, e->rq, e->stream);
    HIPCHK(hipGetLastError());
    e->refine_open = true;
    return VF_OK;
}
int vf_engine_refine_step(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (!e || !e->refine_open) return fail(VF_ERR_INVALID, "no refinement open (vf_engine_refine_begin)");
    vf::launch_refine_step(e->v, e->rq, e->refine_stop(), e->stream);
    HIPCHK(hipGetLastError());
    return VF_OK;
}
int vf_engine_refine_end(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (!e || !e->refine_open) return fail(VF_ERR_INVALID, "no refinement open (vf_engine_refine_begin)");
    vf::launch_refine_end(e->v, e->rq, e->stream);
    HIPCHK(hipGetLastError());
    e->refine_open = false;
    return VF_OK;
}
// corrections the last refined solve of `window` applied before its stopping rule (or the count) ended it; the ratio of the
// preconditioned residual res . M^-1 res at the end to its first value
int vf_engine_read_refine(vf_engine* e, int window, int* corrections, double* reduction) {
    DeviceGuard dev_guard_(e);
    int rc = check_window(e, window);
    if (rc) return rc;
    int it = 0;
    double rz = 0.0, rz0 = 0.0;
    if (e->rq_ready) {
        HIPCHK(hipMemcpyAsync(&it, e->rq.iters + window, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(&rz, e->rq.rz + window, sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(&rz0, e->rq.rz0 + window, sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    if (corrections) *corrections = it;
    if (reduction) *reduction = rz0 > 0.0 ? rz / rz0 : 0.0;
    return VF_OK;
}
int vf_engine_retract(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    vf::launch_retract(e->v, e->stream);
    HIPCHK(hipGetLastError());
    return VF_OK;
}
// non-monotone LM: a failed excursion has put the point it started from back into the current buffer; its factors are
// linearised again (the kernels skip every window whose flag is clear)
static int relinearize_restored(vf_engine* e) {
    vf::View a = e->v;
    a.relin_only = 1;
    if (a.B <= 128 || a.sh_G > 1) vf::launch_linearize_all(a, 0, e->stream);
    else { vf::launch_linearize_imu(a, 0, e->stream); vf::launch_linearize_between_prior(a, 0, e->stream); }
    HIPCHK(hipMemsetAsync(e->v.relin, 0, e->v.B * sizeof(int), e->stream));
    return VF_OK;
}
// ... and an excursion still open when a solve's trials run out is undone (vf_engine_iterate does this itself; callers that
// stage their trials -- time-sharded windows -- call it after the last one)
int vf_engine_close_excursions(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (!e->v.x_best) return VF_OK;
    vf::launch_close_excursions(e->v, e->stream);
    if (int rc = relinearize_restored(e)) return rc;
    // the damping of a window that runs the non-monotone rule persists from solve to solve (k_reset_lambda honours the flag):
    // lambda0 = 1e-5 is six orders above the soft eigenvalues of a window long enough to get here, and a solve of five trials
    // that starts there each time spends them all coming down (measured at 8 000 keyframes: 1.5e-5 m from the optimum after
    // four solves of five trials, against 1e-9 after eight trials in one)
    HIPCHK(hipMemsetAsync(e->v.carry, 0x01, e->v.B * sizeof(int), e->stream));
    HIPCHK(hipGetLastError());
    return VF_OK;
}
int vf_engine_decide(vf_engine* e, int init) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    e->v.nm_W = e->excursion();
    if (e->v.nm_W > 0)
        if (int rc = e->ensure_excursion()) return rc;
    vf::launch_decide(e->v, init ? 1 : 0, e->stream);
    if (e->v.nm_W > 0 && !init)
        if (int rc = relinearize_restored(e)) return rc;
    if (e->hybrid && e->v.stop_on) {          // what the next K4 launch gates on (and the list its sweeps take their windows from)
        vf::View c = e->v;
        c.act = e->act_list;
        vf::launch_count_active(c, e->stream);
    }
    HIPCHK(hipGetLastError());
    return VF_OK;
}
static int iterate_sequence(vf_engine* e, int iterations) {
    // every solve starts from lambda0, as a fresh LevenbergMarquardtOptimizer would (k_reset_lambda: but for a window whose
    // previous solve ended inside an excursion)
    vf::launch_reset_lambda(e->v, e->lambda0_dev, e->stream);
    int rc;
    const int tail = e->slid + e->redo;
    const int slid = (e->warm && e->v.sh_G <= 1 && tail >= 1 && tail <= 8) ? tail : 0;
    if (slid) {
        // nothing but slides since the last solve: only the appended keyframes' factors and the priors need linearising
        HIPCHK(hipMemsetAsync(e->v.done, 0, e->v.B * sizeof(int), e->stream));
        vf::launch_linearize_tail(e->v, slid, e->stream);
        HIPCHK(hipGetLastError());
    } else if ((rc = vf_engine_linearize(e, 0))) return rc;
    if ((rc = vf_engine_decide(e, 1))) return rc;
    // One window under the termination rule (the GraphManager's solve): a trial the rule has made unnecessary is eight launches
    // that do nothing (2-4 us each).  Two trials are enqueued blind -- the rule needs two to see convergence, and that is what
    // a steady update takes --; before each further one the window's flag is read (one small synchronisation, paid only by the
    // solves that go on).  The skipped launches would have skipped the window on the device: same bits.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(e->stream, &capturing);
    const bool adaptive = e->v.B == 1 && e->v.stop_on && e->async_now() && capturing == hipStreamCaptureStatusNone;
    for (int it = 0; it < iterations; it++) {
        if (adaptive && it >= 2) {
            HIPCHK(hipMemcpyAsync(&e->res_host->pad, e->v.done, sizeof(int), hipMemcpyDeviceToHost, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            if (e->res_host->pad) break;
        }
        if ((rc = vf_engine_assemble(e))) return rc;
        if ((rc = vf_engine_solve(e))) return rc;
        if ((rc = vf_engine_retract(e))) return rc;
        if ((rc = vf_engine_linearize(e, 1))) return rc;
        if ((rc = vf_engine_decide(e, 0))) return rc;
    }
    if (iterations > 0 && e->excursion() > 0 && (rc = vf_engine_close_excursions(e))) return rc;
    return VF_OK;
}
// the solve leaves every record, H row and g entry consistent with the current states: the next one may start warm.
// (set by vf_engine_iterate, not by iterate_sequence: a hipGraph replay never runs the sequence's host code)
static void mark_solved(vf_engine* e) {
    e->warm = !e->no_warm && e->x_used == 0;     // (engines holding far between factors start every solve cold)
    e->slid = 0;
    e->redo = 0;
}
int vf_engine_iterate(vf_engine* e, int iterations) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (iterations < 0) return fail(VF_ERR_INVALID, "iterations < 0");
    if (int rc0 = not_sharded(e, "vf_engine_iterate")) return rc0;
    // (asynchronous, like the stages: every read-back synchronises the stream)
    if (e->graph_off || !e->own_stream || e->refine_iters() > 0 || e->excursion() > 0) {
        const int rc = iterate_sequence(e, iterations);
        if (!rc) mark_solved(e);
        return rc;
    }
    const int mode = (e->warm && e->v.sh_G <= 1 && e->slid + e->redo >= 1 && e->slid + e->redo <= 8) ? e->slid + e->redo : 0;
    if (!e->graph_exec || e->graph_iters != iterations || e->graph_epoch != e->epoch || e->graph_mode != mode) {
        e->drop_graph();
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            e->graph_off = true;
            const int rc = iterate_sequence(e, iterations);
            if (!rc) mark_solved(e);
            return rc;
        }
        const int rc = iterate_sequence(e, iterations);
        const hipError_t ce = hipStreamEndCapture(e->stream, &e->graph);
        if (rc) { e->drop_graph(); return rc; }
        if (ce != hipSuccess || hipGraphInstantiate(&e->graph_exec, e->graph, nullptr, nullptr, 0) != hipSuccess) {
            e->drop_graph();
            e->graph_off = true;          // this runtime cannot capture the sequence: plain launches from now on
            (void)hipGetLastError();
            const int rc2 = iterate_sequence(e, iterations);
            if (!rc2) mark_solved(e);
            return rc2;
        }
        e->graph_iters = iterations;
        e->graph_epoch = e->epoch;
        e->graph_mode = mode;
        e->graph_captures++;
    }
    HIPCHK(hipGraphLaunch(e->graph_exec, e->stream));
    e->graph_replays++;
    mark_solved(e);
    return VF_OK;
}
int vf_engine_solve_form(vf_engine* e, int* form) {
    if (!e || !form) return fail(VF_ERR_INVALID, "null argument");
    const vf::View& v = e->v;
    if (v.P >= 2) *form = 4;
    else if (e->hybrid && v.stop_on) *form = 5;
    else if (v.B <= v.tw_max) *form = 3;
    else if (assembles_in_solve(e)) *form = 2;
    else if (v.split_min > 0 && v.B >= v.split_min) *form = 1;
    else *form = 0;
    return VF_OK;
}
int vf_engine_graph_info(vf_engine* e, int* enabled, int* captures, long* replays) {
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (enabled) *enabled = (!e->graph_off && e->own_stream) ? 1 : 0;
    if (captures) *captures = e->graph_captures;
    if (replays) *replays = e->graph_replays;
    return VF_OK;
}

// ------------------------------------------------------------------ chunk geometry (host only, no device needed)
int vf_chunk_geometry(int n, int chunks, int fit, int c, int* count, int* first, int* interior, int* has_separator) {
    if (n < 0 || chunks < 1) return fail(VF_ERR_INVALID, "bad geometry query");
    const int Pe = vf::chunk_count(n, chunks, fit ? 1 : 0);
    if (count) *count = Pe;
    if (c < 0 || c >= Pe) {
        if (first || interior || has_separator) return fail(VF_ERR_INVALID, "chunk %d out of range (%d)", c, Pe);
        return VF_OK;
    }
    const vf::ChunkGeom g = vf::chunk_geom(n, Pe, c);
    if (first) *first = g.i0;
    if (interior) *interior = g.ni;
    if (has_separator) *has_separator = g.has_sep;
    return VF_OK;
}
int vf_shard_range(int n, int chunks, int fit, int rank, int world, int* chunk_lo, int* chunk_hi, int* kf_lo, int* kf_hi) {
    if (n < 0 || chunks < 1 || world < 1 || rank < 0 || rank >= world) return fail(VF_ERR_INVALID, "bad shard query");
    const int Pe = vf::chunk_count(n, chunks, fit ? 1 : 0);
    const int c0 = (int)((long)rank * Pe / world), c1 = (int)((long)(rank + 1) * Pe / world);
    if (chunk_lo) *chunk_lo = c0;
    if (chunk_hi) *chunk_hi = c1;
    if (kf_lo) *kf_lo = c0 < Pe ? vf::chunk_geom(n, Pe, c0).i0 : n;
    if (kf_hi) *kf_hi = c1 < Pe ? vf::chunk_geom(n, Pe, c1).i0 : n;
    return VF_OK;
}

// ------------------------------------------------------------------ time-sharded windows (multi-GPU)
int vf_engine_set_stream(vf_engine* e, void* hip_stream) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->own_stream && e->stream) HIPCHK(hipStreamDestroy(e->stream));
    e->drop_graph();
    e->stream = (hipStream_t)hip_stream;      // nullptr = the device's default stream
    e->own_stream = false;
    if (e->far_columns) return vf_engine_set_stream(e->far_columns, hip_stream);
    return VF_OK;
}
int vf_engine_set_shard(vf_engine* e, int rank, int world) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (world < 1 || rank < 0 || rank >= world) return fail(VF_ERR_INVALID, "bad shard %d of %d", rank, world);
    if (world > 1 && (e->v.P < 2 || e->v.P_fit))
        return fail(VF_ERR_INVALID, "time sharding needs an explicit chunk count (vf_engine_opts.chunks >= 2)");
    if (world > 1 && e->v.P % world != 0) return fail(VF_ERR_INVALID, "chunks (%d) must be a multiple of the world size (%d)", e->v.P, world);
    HIPCHK(hipStreamSynchronize(e->stream));
    e->v.sh_r = rank;
    e->v.sh_G = world;
    e->epoch++;
    return VF_OK;
}
int vf_engine_shard_info(vf_engine* e, vf_shard_info* out) {
    DeviceGuard dev_guard_(e);
    if (!e || !out) return fail(VF_ERR_INVALID, "null argument");
    if (e->v.P < 2) return fail(VF_ERR_INVALID, "engine was not created with the partitioned solve (chunks >= 2)");
    memset(out, 0, sizeof(*out));
    out->rank = e->v.sh_r; out->world = e->v.sh_G; out->windows = e->v.B; out->chunks = e->v.P;
    out->sep = e->v.sepR;
    out->sep_per_chunk = (long)e->v.B * vf::SEPK;
    out->delta = e->v.delta; out->delta_count = e->v.G * 15 + e->v.B;
    // (the refinement's work vectors only for engines that refine as their windows stand NOW: ask again after loading longer ones)
    if (e->refine_iters() > 0) {
        if (int rc = e->ensure_refine()) return rc;
        HIPCHK(hipStreamSynchronize(e->stream));
        out->refine_delta = e->rq.z;
    }
    return VF_OK;
}
static int check_sharded(vf_engine* e) {
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (e->v.P < 2) return fail(VF_ERR_INVALID, "engine was not created with the partitioned solve (chunks >= 2)");
    if (e->v.sh_G > 1)
        for (int w = 0; w < e->v.B; w++) {
            const int n = e->h_hi[w] - e->h_lo[w];
            if (n > 0 && vf::chunk_count(n, e->v.P, e->v.P_fit) != e->v.P)
                return fail(VF_ERR_INVALID, "window %d (%d keyframes) is too short for %d chunks", w, n, e->v.P);
        }
    return VF_OK;
}
int vf_engine_solve_local(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    int rc = check_sharded(e);
    if (rc) return rc;
    vf::View a = e->v;
    if (e->refine_open) { a.gvec = e->rq.nres; a.delta = e->rq.z; a.stop_on = 1; a.done = e->rq.stop; }    // (converged windows are skipped)
    vf::launch_partitioned_local(a, e->stream);
    HIPCHK(hipGetLastError());
    return VF_OK;
}
int vf_engine_solve_global(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    int rc = check_sharded(e);
    if (rc) return rc;
    vf::View a = e->v;
    if (e->refine_open) { a.gvec = e->rq.nres; a.delta = e->rq.z; a.stop_on = 1; a.done = e->rq.stop; }
    vf::launch_partitioned_global(a, e->stream);
    if (e->v.sh_G > 1) vf::launch_mask_delta(a, e->stream);
    HIPCHK(hipGetLastError());
    return VF_OK;
}
// ------------------------------------------------------------------ the collectives of a time-sharded window from C (RCCL)
// A C / C++ caller (the reference's node is C++) cannot reach torch.distributed: here the library issues the two collectives
// of a solve itself, on the engine's stream, through the communicator the caller hands in (ncclComm_t; created by the caller
// with ncclCommInitRank / ncclCommInitAll).  librccl is looked up at run time (dlopen): libvilfusion.so itself keeps no
// communication dependency, and a process that never calls these entry points never loads it.
namespace {
typedef int (*rccl_all_gather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*rccl_all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*rccl_error_string_t)(int);
struct Rccl {
    void* handle = nullptr;
    rccl_all_gather_t all_gather = nullptr;
    rccl_all_reduce_t all_reduce = nullptr;
    rccl_error_string_t error_string = nullptr;
    bool tried = false;
};
Rccl g_rccl;
constexpr int RCCL_FLOAT64 = 8, RCCL_SUM = 0;      // ncclFloat64, ncclSum (rccl.h)
std::once_flag g_rccl_once;
std::string g_rccl_why;
int rccl_load() {
    // (two engines on two threads may ask at once: the library is opened by one of them)
    std::call_once(g_rccl_once, [] {
        g_rccl.tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (g_rccl.handle) {
            g_rccl.all_gather = (rccl_all_gather_t)dlsym(g_rccl.handle, "ncclAllGather");
            g_rccl.all_reduce = (rccl_all_reduce_t)dlsym(g_rccl.handle, "ncclAllReduce");
            g_rccl.error_string = (rccl_error_string_t)dlsym(g_rccl.handle, "ncclGetErrorString");
        }
        if (!g_rccl.all_gather || !g_rccl.all_reduce) {
            const char* why = dlerror();
            g_rccl_why = why ? why : "symbols missing";
        }
    });
    if (!g_rccl.all_gather || !g_rccl.all_reduce)
        return fail(VF_ERR_DEVICE, "librccl not found (dlopen librccl.so.1): %s", g_rccl_why.c_str());
    return VF_OK;
}
int rccl_check(int rc, const char* what) {
    if (rc == 0) return VF_OK;
    return fail(VF_ERR_DEVICE, "%s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "rccl error");
}
// one staged solve of the shard: local phase, all-gather of the packed separator system (in place: the rank's own slice of
// the receive buffer is its send buffer), global phase, all-reduce of increments + failure flags; then the refinement, each
// correction the same two collectives on the correction's buffers
int shard_solve(vf_engine* e, void* comm) {
    int rc;
    const vf::View& v = e->v;
    const size_t per_chunk = (size_t)v.B * vf::SEPK, per_rank = (size_t)(v.P / v.sh_G) * per_chunk;
    const size_t nd = (size_t)v.G * 15 + (size_t)v.B;
    auto exchange = [&](double* delta) -> int {
        int r;
        if ((r = vf_engine_solve_local(e))) return r;
        if ((r = rccl_check(g_rccl.all_gather(v.sepR + (size_t)v.sh_r * per_rank, v.sepR, per_rank, RCCL_FLOAT64, comm, e->stream), "ncclAllGather"))) return r;
        if ((r = vf_engine_solve_global(e))) return r;
        return rccl_check(g_rccl.all_reduce(delta, delta, nd, RCCL_FLOAT64, RCCL_SUM, comm, e->stream), "ncclAllReduce");
    };
    if ((rc = vf_engine_assemble(e)) || (rc = exchange(e->v.delta))) return rc;
    int R = 0;
    if ((rc = vf_engine_refine_count(e, &R))) return rc;
    if (R > 0) {
        if ((rc = vf_engine_refine_begin(e))) return rc;
        for (int it = 0; it < R; it++)
            if ((rc = exchange(e->rq.z)) || (rc = vf_engine_refine_step(e))) {
                e->refine_open = false;       // (a failed collective must not leave later solves on the correction's buffers)
                return rc;
            }
        if ((rc = vf_engine_refine_end(e))) { e->refine_open = false; return rc; }
    }
    return VF_OK;
}
int shard_ready(vf_engine* e, void* comm, const char* what) {
    if (!e || !comm) return fail(VF_ERR_INVALID, "%s: null argument", what);
    if (e->v.P < 2 || e->v.P_fit) return fail(VF_ERR_INVALID, "%s: the engine needs an explicit chunk count (vf_engine_opts.chunks >= 2) and vf_engine_set_shard", what);
    return rccl_load();
}
}  // namespace

int vf_shard_exchange_plan(int windows, int capacity, int chunks, int rank, int world, long* sep_offset, long* sep_count,
                           long* sep_total, long* delta_count) {
    if (windows < 1 || capacity < 1 || chunks < 2 || world < 1 || rank < 0 || rank >= world || chunks % world != 0)
        return fail(VF_ERR_INVALID, "bad exchange-plan query");
    const long M = (capacity + 63) / 64 * 64, per_chunk = (long)windows * vf::SEPK, per_rank = (long)(chunks / world) * per_chunk;
    if (sep_offset) *sep_offset = rank * per_rank;
    if (sep_count) *sep_count = per_rank;
    if (sep_total) *sep_total = (long)chunks * per_chunk;
    if (delta_count) *delta_count = (long)windows * M * 15 + windows;
    return VF_OK;
}
int vf_shard_iterate(vf_engine* e, void* nccl_comm, int iterations) {
    DeviceGuard dev_guard_(e);
    int rc = shard_ready(e, nccl_comm, "vf_shard_iterate");
    if (rc) return rc;
    if (iterations < 0) return fail(VF_ERR_INVALID, "iterations < 0");
    if ((rc = vf_engine_reset_lambda(e)) || (rc = vf_engine_linearize(e, 0)) || (rc = vf_engine_decide(e, 1))) return rc;
    for (int it = 0; it < iterations; it++)
        if ((rc = shard_solve(e, nccl_comm)) || (rc = vf_engine_retract(e)) || (rc = vf_engine_linearize(e, 1)) || (rc = vf_engine_decide(e, 0))) return rc;
    if (iterations > 0 && (rc = vf_engine_close_excursions(e))) return rc;
    return VF_OK;
}
int vf_shard_gn_step(vf_engine* e, void* nccl_comm, double relin_threshold) {
    DeviceGuard dev_guard_(e);
    int rc = shard_ready(e, nccl_comm, "vf_shard_gn_step");
    if (rc) return rc;
    if ((rc = vf_engine_gn_begin(e, relin_threshold)) || (rc = shard_solve(e, nccl_comm)) || (rc = vf_engine_retract(e))) return rc;
    return VF_OK;
}

int vf_engine_set_convergence(vf_engine* e, double rel_tol, double abs_tol) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (!(rel_tol >= 0.0) || !(abs_tol >= 0.0)) return fail(VF_ERR_INVALID, "tolerances must be >= 0");
    HIPCHK(hipStreamSynchronize(e->stream));
    e->v.rel_tol = rel_tol;
    e->v.abs_tol = abs_tol;
    e->v.stop_on = (rel_tol > 0.0 || abs_tol > 0.0) ? 1 : 0;
    e->epoch++;
    // sweep engines (large batches): once few windows are left taking trials, K4 switches to the partitioned form
    if (e->v.stop_on && e->v.P == 0 && e->v.B > 128 && !e->hybrid && e->tune.hybrid_threshold >= 0) {
        vf::View& v = e->v;
        e->hybrid_P = vf::chunk_count(v.M, 96, 1);
        if (e->hybrid_P >= 2) {
            const size_t BP = (size_t)v.B * e->hybrid_P;
            int rc;
            if ((rc = e->alloc(&e->h_Vp, (size_t)v.G * vf::VROW)) || (rc = e->alloc(&e->h_sep, BP * vf::SEPK)) ||
                (rc = e->alloc(&e->h_sepL, BP * vf::SEPL)) || (e->tune.hybrid_active_list && (rc = e->alloc(&e->act_list, (size_t)v.B)))) return rc;
            e->hybrid = true;
        }
    }
    HIPCHK(hipMemsetAsync(e->v.done, 0, e->v.B * sizeof(int), e->stream));
    return VF_OK;
}
int vf_engine_reset_lambda(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    vf::launch_reset_lambda(e->v, e->lambda0_dev, e->stream);
    HIPCHK(hipGetLastError());
    return VF_OK;
}

int vf_engine_predict(vf_engine* e, int window, int k0, int n) {
    DeviceGuard dev_guard_(e, e && window >= 0 && window < e->v.B && k0 > e->h_lo[window] + 1);
    touch(e, window, k0);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (window >= e->v.B || k0 < 1 || n < 0 || k0 + n > e->v.M) return fail(VF_ERR_BAD_KEY, "bad predict range");
    if (n == 0) return VF_OK;
    vf::launch_predict(e->v, window, k0, n, 0, e->stream);
    HIPCHK(hipGetLastError());
    return VF_OK;
}

// ------------------------------------------------------------------ reference-compat solve
// What the reference does per GraphManager::solve (GraphManager.cpp:38-43,126-127): ONE iSAM2 update -- Gauss-Newton
// (QR, no damping) about per-variable linearisation points theta, relinearising only the variables whose pending
// increment reaches relinearizeThreshold (1e-4) -- and calculateEstimate() = theta (+) delta.  Here: buffer sel = theta,
// v.delta = delta, the trial buffer = the estimate.  iSAM2 re-eliminates only the cliques a new factor touches; solving the
// whole banded system with every factor linearised at theta gives the same increment (its partial back-substitution
// stops below wildfireThreshold 1e-3 * ... of change; that approximation is NOT reproduced -- the full solve is exact).
// the opening of such an update, also for time-sharded engines (the caller then runs the staged solve -- vf_engine_assemble,
// vf_engine_solve_local, ..., the refinement -- and vf_engine_retract): relinearise where the pending increment reaches the
// threshold, lambda := 0 (Gauss-Newton), linearise every factor at theta
int vf_engine_gn_begin(vf_engine* e, double relin_threshold) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (!(relin_threshold >= 0.0)) return fail(VF_ERR_INVALID, "relinearisation threshold must be >= 0");
    cold(e);
    int rc;
    vf::launch_relinearize(e->v, relin_threshold, e->stream);
    HIPCHK(hipMemsetAsync(e->v.lambda, 0, e->v.B * sizeof(double), e->stream));     // Gauss-Newton: no damping
    if ((rc = vf_engine_linearize(e, 0)) || (rc = vf_engine_decide(e, 1))) return rc;
    return VF_OK;
}
// The same update done incrementally (vf_engine_opts.incremental): relinearise, then linearise / assemble / eliminate only from
// the first keyframe that changed, back-substitute until the increments stop changing (vf_kernels.hpp "Incremental
// Gauss-Newton updates").  The first update of an engine, and every update after an entry point the bookkeeping does not
// follow, covers the whole window.
static int isam_step_incremental(vf_engine* e, double relin_threshold) {
    const int invalid = (e->inc_valid && e->opts.incremental != 2) ? 0 : 1;     // (incremental = 2: the same kernels over the whole window, every time)
    int appended = e->inc_slid;
    if (e->v.B == 1 && e->inc_first_dirty != 0x7fffffff) appended = std::max(appended, e->h_hi[0] - e->inc_first_dirty);
    vf::View a = e->v;
    a.inc_on = 1;
    a.inc_prior = (invalid || e->inc_slid > 0) ? 1 : 0;
    a.stop_on = 0;
    vf::launch_inc_begin(a, relin_threshold, appended, invalid, e->stream);
    HIPCHK(hipMemsetAsync(e->v.lambda, 0, e->v.B * sizeof(double), e->stream));     // Gauss-Newton: no damping
    if (a.B <= 128) vf::launch_linearize_all(a, 0, e->stream);
    else { vf::launch_linearize_imu(a, 0, e->stream); vf::launch_linearize_between_prior(a, 0, e->stream); }
    vf::launch_assemble(a, e->stream);
    vf::launch_inc_solve(a, e->stream);
    vf::launch_inc_retract(a, e->stream);
    HIPCHK(hipGetLastError());
    cold(e);
    e->inc_slid = 0;
    e->inc_first_dirty = 0x7fffffff;
    e->inc_updates++;
    if (invalid) e->inc_full++;
    std::vector<int> failed((size_t)e->v.B);
    HIPCHK(hipMemcpyAsync(failed.data(), e->v.fail, failed.size() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int w = 0; w < e->v.B; w++)
        if (failed[w]) return fail(VF_ERR_INDETERMINATE, "window %d: normal equations not positive definite (underdetermined graph)", w);
    e->inc_valid = true;
    return VF_OK;
}
int vf_engine_isam_step(vf_engine* e, double relin_threshold) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (int rc0 = not_sharded(e, "vf_engine_isam_step")) return rc0;
    if (!(relin_threshold >= 0.0)) return fail(VF_ERR_INVALID, "relinearisation threshold must be >= 0");
    // (far factors are solved as a low-rank correction of the whole band: such windows take the full update)
    if (e->v.ck && e->x_used == 0 && e->refine_iters() == 0) return isam_step_incremental(e, relin_threshold);
    int rc;
    if ((rc = vf_engine_gn_begin(e, relin_threshold)) || (rc = vf_engine_assemble(e)) ||
        (rc = vf_engine_solve(e)) || (rc = vf_engine_retract(e))) return rc;
    std::vector<int> failed((size_t)e->v.B);
    HIPCHK(hipMemcpyAsync(failed.data(), e->v.fail, failed.size() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int w = 0; w < e->v.B; w++)
        if (failed[w]) return fail(VF_ERR_INDETERMINATE, "window %d: normal equations not positive definite (underdetermined graph)", w);
    cold(e);
    return VF_OK;
}
int vf_engine_incremental_info(vf_engine* e, int window, long* updates, long* whole_window_updates, int* first_eliminated, int* last_substituted) {
    DeviceGuard dev_guard_(e);
    int rc = check_window(e, window);
    if (rc) return rc;
    if (updates) *updates = e->inc_updates;
    if (whole_window_updates) *whole_window_updates = e->inc_full;
    int from = -1, stop = -1;
    if (e->v.ck) {
        HIPCHK(hipMemcpyAsync(&from, e->v.inc_from + window, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(&stop, e->v.inc_stop + window, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    if (first_eliminated) *first_eliminated = from;
    if (last_substituted) *last_substituted = stop;
    return VF_OK;
}
int vf_engine_predict_from_estimate(vf_engine* e, int window, int k0, int n) {
    DeviceGuard dev_guard_(e, e && window >= 0 && window < e->v.B && k0 > e->h_lo[window] + 1);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    { const bool keep = e->inc_valid; touch(e, window, k0); cold(e); if (keep && e->v.B == 1 && window == 0 && k0 > e->h_lo[0]) e->inc_valid = true; }
    if (window >= e->v.B || k0 < 1 || n < 0 || k0 + n > e->v.M) return fail(VF_ERR_BAD_KEY, "bad predict range");
    if (n == 0) return VF_OK;
    vf::launch_predict(e->v, window, k0, n, 1, e->stream);
    HIPCHK(hipGetLastError());
    return VF_OK;
}
int vf_engine_get_estimate(vf_engine* e, int window, int k0, int n, double* s) {
    DeviceGuard dev_guard_(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (!s) return fail(VF_ERR_INVALID, "null states");
    if (n == 0) return VF_OK;
    const size_t bytes = (size_t)n * 16 * sizeof(double);
    if ((rc = e->ensure_stage(bytes))) return rc;
    vf::launch_gather_states(e->v.x, e->stage, e->v.G, e->v.sel, e->v.M, 1, (long)window * e->v.M + k0, n, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s, e->stage, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

int vf_engine_marginalize(vf_engine* e) {
    DeviceGuard dev_guard_(e, e && e->async_now());
    // (reads the current linearisation, writes the marginal prior: what a warm start expects to have changed)
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (int rc = not_sharded(e, "vf_engine_marginalize")) return rc;
    for (int w = 0; w < e->v.B; w++)
        if (e->h_hi[w] - e->h_lo[w] < 4) return fail(VF_ERR_INVALID, "window %d: marginalisation needs >= 4 keyframes", w);
    if (e->async_now() && e->ahead_valid && e->ahead_lo == e->h_lo[0] && !e->side_open) {
        // computed behind the previous solve (vf_engine_marginalize_ahead): put in place, nothing to wait for
        vf::launch_marg_commit(e->v, e->marg_stash, e->stream);
        HIPCHK(hipGetLastError());
        e->ahead_valid = false;
        e->ahead_used++;
        e->marg_since_drop = true;
        return VF_OK;
    }
    if (e->async_now()) {
        // beside whatever the main stream is given next (K0, prediction and staging of the keyframe that arrives): a second
        // stream that starts where the main one stands now
        if (!e->side_open) {
            HIPCHK(hipEventRecord(e->ev_fork, e->stream));
            HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
            e->side_open = true;
        }
        vf::launch_marginalize(e->v, e->sticky_dev + 1, e->stream2);
        HIPCHK(hipGetLastError());
        e->marg_since_drop = true;
        return VF_OK;
    }
    HIPCHK(hipMemsetAsync(e->status_dev, 0, sizeof(int), e->stream));
    // far factors the prior is about to absorb: their linearisation at the current states (a compaction, a transport or any
    // other re-sending of the list since the last solve has zeroed the buffers)
    if (e->x_used > 0) {
        vf::View a = e->v;
        a.stop_on = 0;       // (a window the termination rule finished in the last solve still has its `done` flag up: linearise it all the same)
        vf::launch_linearize_extra(a, 0, e->stream);
    }
    vf::launch_marginalize(e->v, e->status_dev, e->stream);
    HIPCHK(hipGetLastError());
    int status = 0;
    HIPCHK(hipMemcpyAsync(&status, e->status_dev, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (status & 4) return fail(VF_ERR_INDETERMINATE, "marginalisation: the far ends' block of the marginal is not positive definite");
    if (status) return fail(VF_ERR_INDETERMINATE, "marginalisation: pivot block of the oldest keyframe not positive definite");
    e->marg_since_drop = true;
    return VF_OK;
}

// ---- far between factors across a slide: the host's side of what k_marginalize<FAR> has just done on the device.  A far
// factor (a -> b) whose older keyframe a has been MARGINALISED was marginalised with it: it is part of the window's linear far
// factor now (View::xl_*), or -- with its far end within the new prior's reach -- inside the prior; the lists here are host
// mirrors and apply the same rule in the same order (survivors of the linear list, then the nonlinear entries anchored at the
// leaving keyframe, by index).  The information of a loop closure thereby outlives the keyframe it was anchored on (iSAM2
// keeps every factor for good, GraphManager.cpp:83-88).
// A slide that does NOT marginalise (re-anchoring: the keyframe is dropped, its band factors with it) re-anchors a nonlinear
// far factor on the next keyframe instead: with D = T_a^-1 T_a+1 at the current estimate, the measurement Z of T_a^-1 T_b becomes
// Z' = D^-1 Z of T_a+1^-1 T_b -- the same residual Log(Z^-1 T_a^-1 T_b), in the same tangent frame (at b), so the square-root
// information stays as it is; D is known from the IMU factor between the two keyframes to ~2e-5 m / 2e-4 rad, against the
// 1e-2 ... 0.5 m a between factor claims, and is taken as exact (two states are read back per window that has such a factor).
// The linear rows, expressed around the prior that such a slide discards, end there.
namespace {
void quat_to_rot_(const double* q, double* R) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
void quat_mul_(const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
}  // namespace
static int transport_far(vf_engine* e, bool marginalised) {
    if (e->x_used == 0) return VF_OK;
    for (int w = 0; w < e->v.B; w++) {
        const int lo = e->h_lo[w], hi = e->h_hi[w];
        bool any = !e->h_lb[w].empty();
        for (int i = 0; i < e->h_xn[w]; i++) any = any || e->h_xa[w][i] == lo;
        if (!any || hi - lo < 2) continue;
        // (1) the linear far factors -- all of them have the leaving keyframe in their support.  k_marginalize has absorbed the
        // ones that end at lo + 3 and re-expressed the others over the next three keyframes, in this order; without a
        // marginalisation (re-anchoring slide) they go the way the band factors of the dropped keyframe go.
        std::vector<int> lb;
        for (int kb : e->h_lb[w]) {
            if (!marginalised) e->far_ended++;
            else if (kb == lo + 3) e->far_absorbed++;
            else lb.push_back(kb);
        }
        const bool had_linear = !e->h_lb[w].empty();
        // (2) the nonlinear ones anchored at lo
        double st[32], R0[9], qD[4], tD[3], RD[9], qDc[4];
        bool have_D = false;
        std::vector<int> a, b;
        std::vector<double> rec;
        for (int i = 0; i < e->h_xn[w]; i++) {
            double r[vf::BTW_IN];
            memcpy(r, e->h_xrec[w].data() + (size_t)i * vf::BTW_IN, sizeof(r));
            int ai = e->h_xa[w][i];
            const int kb = e->h_xb[w][i];
            if (ai == lo) {
                // within the marginal prior's reach (ends at lo+1 .. lo+3): k_marginalize has just absorbed it, like a band
                // factor -- its information lives on in the prior
                if (kb - lo <= 3) { if (marginalised) e->far_absorbed++; else e->far_ended++; continue; }
                // marginalised with its far end inside the window: k_marginalize has made it a linear far factor (exact
                // at the current linearisation), behind the survivors of (1)
                if (marginalised && kb < hi) { lb.push_back(kb); e->far_transported++; continue; }
                // no marginalisation (or an end the window has not reached yet): re-anchored on the next keyframe, the step
                // between the two taken as exact
                if (!have_D) {
                    if (int rc = vf_engine_get_states(e, w, lo, 2, st)) return rc;
                    double qc[4] = {st[0], -st[1], -st[2], -st[3]}, dt[3] = {st[20] - st[4], st[21] - st[5], st[22] - st[6]};     // D = T_lo^-1 T_lo+1
                    quat_to_rot_(st, R0);
                    quat_mul_(qc, st + 16, qD);
                    for (int c = 0; c < 3; c++) tD[c] = R0[0 * 3 + c] * dt[0] + R0[1 * 3 + c] * dt[1] + R0[2 * 3 + c] * dt[2];
                    quat_to_rot_(qD, RD);
                    qDc[0] = qD[0]; qDc[1] = -qD[1]; qDc[2] = -qD[2]; qDc[3] = -qD[3];
                    have_D = true;
                }
                double q2[4], d[3] = {r[4] - tD[0], r[5] - tD[1], r[6] - tD[2]};
                quat_mul_(qDc, r, q2);                                          // R' = R_D^T R_Z
                const double nq = std::sqrt(q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2] + q2[3] * q2[3]);
                for (int c = 0; c < 4; c++) r[c] = q2[c] / nq;
                for (int c = 0; c < 3; c++) r[4 + c] = RD[0 * 3 + c] * d[0] + RD[1 * 3 + c] * d[1] + RD[2 * 3 + c] * d[2];   // t' = R_D^T (t_Z - t_D)
                ai = lo + 1;
                e->far_transported++;
            }
            a.push_back(ai);
            b.push_back(kb);
            rec.insert(rec.end(), r, r + vf::BTW_IN);
        }
        e->h_lb[w] = lb;
        if (had_linear && !marginalised) HIPCHK(hipMemsetAsync(e->v.xl_n + w, 0, sizeof(int), e->stream));
        if (int rc = vf_engine_set_extra_between(e, w, (int)a.size(), a.data(), b.data(), rec.data())) return rc;
    }
    return VF_OK;
}
int vf_engine_get_linear_far(vf_engine* e, int window, int* n, int32_t* far_end) {
    int rc = check_window(e, window);
    if (rc) return rc;
    const int cnt = e->v.x_max ? e->h_ln(window) : 0;
    if (n) *n = cnt;
    for (int i = 0; i < cnt && far_end; i++) far_end[i] = e->h_lb[window][i];
    return VF_OK;
}
int vf_engine_get_extra_between(vf_engine* e, int window, int* n, int32_t* a, int32_t* b, double* rec28, long* transported, long* ended, long* absorbed) {
    int rc = check_window(e, window);
    if (rc) return rc;
    const int cnt = e->v.x_max ? e->h_xn[window] : 0;
    if (n) *n = cnt;
    for (int i = 0; i < cnt; i++) {
        if (a) a[i] = e->h_xa[window][i];
        if (b) b[i] = e->h_xb[window][i];
        if (rec28) memcpy(rec28 + (size_t)i * vf::BTW_IN, e->h_xrec[window].data() + (size_t)i * vf::BTW_IN, sizeof(double) * vf::BTW_IN);
    }
    if (transported) *transported = e->far_transported;
    if (ended) *ended = e->far_ended;
    if (absorbed) *absorbed = e->far_absorbed;
    return VF_OK;
}

int vf_engine_marginalize_ahead(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    // (a warm engine: its records are those of the current states -- what a marginalisation reads -- and stay so while only
    // appends follow; anything else voids the stash through cold())
    if (!e->async_now() || !e->warm || e->h_hi[0] - e->h_lo[0] < 4) return VF_OK;
    if (int rc = e->ensure_async()) return rc;
    if (!e->marg_stash) {
        HIPCHK(hipMalloc((void**)&e->marg_stash, (size_t)e->v.B * vf::MARG_STASH_DOUBLES * sizeof(double)));
        e->allocs.push_back(e->marg_stash);
    }
    vf::launch_marginalize_ahead(e->v, e->sticky_dev + 1, e->marg_stash, e->stream);
    HIPCHK(hipGetLastError());
    e->ahead_valid = true;
    e->ahead_lo = e->h_lo[0];
    e->ahead_made++;
    return VF_OK;
}
int vf_engine_drop_oldest(vf_engine* e) {
    DeviceGuard dev_guard_(e, e && e->async_now());
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (e->async_now()) {
        for (int w = 0; w < e->v.B; w++)
            if (e->h_hi[w] - e->h_lo[w] < 2) return fail(VF_ERR_INVALID, "window %d too short", w);
        vf::launch_bump_lo(e->v, e->side_open ? e->stream2 : e->stream);       // (behind the marginalisation it follows, wherever that runs)
        HIPCHK(hipGetLastError());
        e->marg_since_drop = false;
        for (int w = 0; w < e->v.B; w++) e->h_lo[w]++;
        return VF_OK;
    }
    if (int rc = transport_far(e, e->marg_since_drop)) return rc;
    e->marg_since_drop = false;
    for (int w = 0; w < e->v.B; w++) {
        if (e->h_hi[w] - e->h_lo[w] < 2) return fail(VF_ERR_INVALID, "window %d too short", w);
        const int lo = e->h_lo[w] + 1;
        HIPCHK(hipMemcpyAsync(e->v.lo + w, &lo, sizeof(int), hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int w = 0; w < e->v.B; w++) e->h_lo[w]++;
    return VF_OK;
}

int vf_engine_slide(vf_engine* e, const double* prior_sigma15, int marginalize) {
    DeviceGuard dev_guard_(e);
    if (!e || !prior_sigma15) return fail(VF_ERR_INVALID, "null argument");
    for (int w = 0; w < e->v.B; w++)
        if (e->h_hi[w] >= e->v.M) return fail(VF_ERR_CAPACITY, "window %d has no free keyframe slot", w);
    if (marginalize) {
        int rc = vf_engine_marginalize(e);
        if (rc) return rc;
    }
    if (int rc = transport_far(e, marginalize != 0)) return rc;
    e->marg_since_drop = false;
    if (e->warm) e->slid++;   // a slide is a change a warm start knows how to follow
    if (e->inc_valid) e->inc_slid++;
    // the sigmas are a caller temporary: uploaded (and waited for) only when they differ from what the device already holds, so
    // that a run of updates with the same sigmas -- every fixed-lag loop -- enqueues without a host synchronisation
    const bool fresh_sigma = !e->sigma_valid || memcmp(e->sigma_host, prior_sigma15, sizeof(e->sigma_host)) != 0;
    if (fresh_sigma) {
        HIPCHK(hipMemcpyAsync(e->sigma_dev, prior_sigma15, 15 * sizeof(double), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        memcpy(e->sigma_host, prior_sigma15, sizeof(e->sigma_host));
        e->sigma_valid = true;
    }
    vf::launch_slide(e->v, e->sigma_dev, marginalize ? 0 : 1, e->stream);
    HIPCHK(hipGetLastError());
    for (int w = 0; w < e->v.B; w++) { e->h_lo[w]++; e->h_hi[w]++; }
    return VF_OK;
}

int vf_engine_read_marginal(vf_engine* e, int window, int* on, double* xbar48, double* L729, double* eta27) {
    DeviceGuard dev_guard_(e);
    int rc = check_window(e, window);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (on) HIPCHK(hipMemcpy(on, e->v.mp_on + window, sizeof(int), hipMemcpyDeviceToHost));
    if (xbar48) HIPCHK(hipMemcpy(xbar48, e->v.mp_x + (size_t)window * 48, 48 * sizeof(double), hipMemcpyDeviceToHost));
    if (L729) HIPCHK(hipMemcpy(L729, e->v.mp_L + (size_t)window * 729, 729 * sizeof(double), hipMemcpyDeviceToHost));
    if (eta27) HIPCHK(hipMemcpy(eta27, e->v.mp_eta + (size_t)window * 27, 27 * sizeof(double), hipMemcpyDeviceToHost));
    return VF_OK;
}

// Move the live keyframes [shift, M) of every window to [0, M - shift): frees `shift` slots at the
// end.  shift must be a multiple of 64 (whole AoSoA tiles) and <= every window's lo.
int vf_engine_compact(vf_engine* e, int shift) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    vf::View& v = e->v;
    if (shift <= 0 || shift % 64 != 0 || shift >= v.M) return fail(VF_ERR_INVALID, "shift must be a positive multiple of 64 below the capacity");
    for (int w = 0; w < v.B; w++)
        if (e->h_lo[w] < shift) return fail(VF_ERR_BAD_KEY, "window %d: lo %d < shift %d (live keyframes would be lost)", w, e->h_lo[w], shift);
    const size_t keepk = (size_t)(v.M - shift);              // slots kept per window
    // staging buffer: the largest per-window segment moved at once (imu_out tiles)
    constexpr int JNF = vf::JT_STRIDE * (64 / vf::JT) / 64;      // the J stream of 64 slots, counted in "fields" of 64 doubles (292)
    static_assert(JNF * 64 == vf::JT_STRIDE * (64 / vf::JT), "J stream tiles per 64 slots");
    const size_t seg_max = (keepk / 64) * (size_t)(JNF > vf::IMU_IN ? JNF : vf::IMU_IN) * 64;
    int rc = e->ensure_stage(seg_max * sizeof(double));
    if (rc) return rc;
    auto move = [&](double* base, size_t per_slot_window_stride, size_t src_off, size_t n) -> int {
        // base + w*stride + src_off  ->  base + w*stride, n doubles, via the staging buffer
        for (int w = 0; w < v.B; w++) {
            double* dst = base + (size_t)w * per_slot_window_stride;
            vf::launch_shift_copy(dst + src_off, e->stage, (long)n, e->stream);
            vf::launch_shift_copy(e->stage, dst, (long)n, e->stream);
        }
        HIPCHK(hipGetLastError());
        return VF_OK;
    };
    const size_t G = (size_t)v.G, tilesG = G / 64, tilesM = (size_t)v.M / 64, tshift = (size_t)shift / 64;
    // states: [2][16][G] -> per (buf, comp) plane, per window a run of M doubles
    for (int bc = 0; bc < 32; bc++)
        if ((rc = move(v.x + (size_t)bc * G, (size_t)v.M, (size_t)shift, keepk))) return rc;
    // AoSoA arrays: per window tilesM tiles of nf*64 doubles
    struct { double* p; int nf; int bufs; } arrs[] = {{v.imu_in, vf::IMU_IN, 1}, {v.imu_r, vf::IMU_R, 2}, {v.imu_j, JNF, 2},
                                                       {v.btw_in, vf::BTW_IN, 1}, {v.btw_out, vf::BTW_OUT, 2}};
    for (auto& a : arrs)
        for (int bf = 0; bf < a.bufs; bf++) {
            const size_t tile = (size_t)a.nf * 64;
            if ((rc = move(a.p + (size_t)bf * tilesG * tile, tilesM * tile, tshift * tile, (tilesM - tshift) * tile))) return rc;
        }
    // between-factor source indices: move (as raw 4-byte ints, staged through the double buffer) then rebase
    {
        int* stage_i = (int*)e->stage;
        for (int w = 0; w < v.B; w++) {
            int* dst = v.btw_a + (size_t)w * v.M;
            HIPCHK(hipMemcpyAsync(stage_i, dst + shift, keepk * sizeof(int), hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(dst, stage_i, keepk * sizeof(int), hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipMemsetAsync(dst + keepk, 0xff, (size_t)shift * sizeof(int), e->stream));
        }
        vf::launch_shift_btw_a(v.btw_a, v.G, v.M, shift, e->stream);
        HIPCHK(hipGetLastError());
    }
    // window ranges and prior keys
    std::vector<int> lo(v.B), hi(v.B), pk(v.B);
    HIPCHK(hipMemcpyAsync(pk.data(), v.prior_k, v.B * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int w = 0; w < v.B; w++) {
        e->h_lo[w] -= shift; e->h_hi[w] -= shift;
        lo[w] = e->h_lo[w]; hi[w] = e->h_hi[w];
        if (pk[w] >= 0) pk[w] -= shift;
    }
    HIPCHK(hipMemcpyAsync(v.lo, lo.data(), v.B * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(v.hi, hi.data(), v.B * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(v.prior_k, pk.data(), v.B * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    // far between factors: slots move down with the keyframes; one whose older keyframe was reclaimed is gone
    for (int w = 0; w < v.B && v.x_max > 0; w++) {
        if (e->h_xn[w] == 0) continue;
        std::vector<int> a, b;
        std::vector<double> r;
        for (int i = 0; i < e->h_xn[w]; i++) {
            if (e->h_xa[w][i] < shift) continue;
            a.push_back(e->h_xa[w][i] - shift);
            b.push_back(e->h_xb[w][i] - shift);
            r.insert(r.end(), e->h_xrec[w].begin() + (size_t)i * vf::BTW_IN, e->h_xrec[w].begin() + (size_t)(i + 1) * vf::BTW_IN);
        }
        if (int rc = vf_engine_set_extra_between(e, w, (int)a.size(), a.data(), b.data(), r.data())) return rc;
    }
    for (int w = 0; w < v.B && v.x_max > 0; w++) {          // ... and so do the far ends of the linear ones (always inside the window)
        if (e->h_lb[w].empty()) continue;
        for (int& kb : e->h_lb[w]) kb -= shift;
        HIPCHK(hipMemcpyAsync(v.xl_b + (size_t)w * v.x_max, e->h_lb[w].data(), e->h_lb[w].size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return VF_OK;
}

// New engine with more keyframe slots per window, the problem carried over on the device, then swapped into *e (the
// handle the caller holds stays valid).  Inputs and states only: every linearisation, H, g and panel is recomputed by
// the next (cold) solve.
int vf_engine_grow(vf_engine* e, int new_capacity) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    if (int rc0 = not_sharded(e, "vf_engine_grow")) return rc0;
    const int M0 = e->v.M, M1 = (new_capacity + 63) / 64 * 64;
    if (M1 <= M0) return fail(VF_ERR_INVALID, "new capacity %d does not exceed the current %d", new_capacity, M0);
    vf_engine_opts o = e->opts;
    o.capacity = M1;
    vf_engine* n = nullptr;
    int rc = create_engine(&o, &e->tune, true, e->stream, &n);     // (on the stream the handle has: the old engine's, handed over below)
    if (rc) return rc;
    const vf::View &a = e->v, &b = n->v;
    const int B = a.B;
    const size_t G0 = (size_t)a.G, G1 = (size_t)b.G;
    // every copy through one lambda that remembers the first failure: the new engine is destroyed on any error path
    hipError_t herr = hipStreamSynchronize(e->stream);
    auto cp = [&](void* dst, const void* src, size_t bytes) {
        if (herr == hipSuccess) herr = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, n->stream);
    };
    // a per-window array with `unit` bytes per keyframe slot is B rows of M0 * unit bytes, pitch M0 -> M1 slots: ONE pitched copy
    // per array (36 calls whatever the number of windows; the first version issued 36 copies per window)
    auto cp2 = [&](void* dst, const void* src, size_t unit) {
        if (herr == hipSuccess)
            herr = hipMemcpy2DAsync(dst, (size_t)M1 * unit, src, (size_t)M0 * unit, (size_t)M0 * unit, (size_t)B, hipMemcpyDeviceToDevice, n->stream);
    };
    for (int pl = 0; pl < 32; pl++) cp2(b.x + (size_t)pl * G1, a.x + (size_t)pl * G0, sizeof(double));     // state planes [2][16][G]
    cp2(b.imu_in, a.imu_in, vf::IMU_IN * sizeof(double));        // AoSoA tiles of 64 slots: whole tiles move (M0, M1 are multiples of 64)
    cp2(b.btw_in, a.btw_in, vf::BTW_IN * sizeof(double));
    cp2(b.btw_a, a.btw_a, sizeof(int));
    cp2(b.delta, a.delta, 15 * sizeof(double));
    struct { void* d; const void* s; size_t bytes; } per_window[] = {
        {b.prior_k, a.prior_k, B * sizeof(int)}, {b.prior_in, a.prior_in, (size_t)B * vf::PRIOR_IN * sizeof(double)},
        {b.mp_on, a.mp_on, B * sizeof(int)}, {b.mp_x, a.mp_x, (size_t)B * 48 * sizeof(double)}, {b.mp_L, a.mp_L, (size_t)B * 729 * sizeof(double)},
        {b.mp_eta, a.mp_eta, (size_t)B * 27 * sizeof(double)}, {b.lo, a.lo, B * sizeof(int)}, {b.hi, a.hi, B * sizeof(int)},
        {b.sel, a.sel, B * sizeof(int)}, {b.lambda, a.lambda, B * sizeof(double)}, {b.cost, a.cost, B * sizeof(double)},
        {b.n_acc, a.n_acc, B * sizeof(int)}, {b.n_rej, a.n_rej, B * sizeof(int)}, {b.n_fail, a.n_fail, B * sizeof(int)},
        {n->lambda0_dev, e->lambda0_dev, B * sizeof(double)}};
    for (auto& c : per_window) cp(c.d, c.s, c.bytes);
    if (herr == hipSuccess) herr = hipStreamSynchronize(n->stream);
    if (herr != hipSuccess) {
        vf_engine_destroy(n);
        return fail(VF_ERR_DEVICE, "vf_engine_grow: device copy failed: %s", hipGetErrorString(herr));
    }
    n->h_lo = e->h_lo;
    n->h_hi = e->h_hi;
    if (e->v.stop_on && (rc = vf_engine_set_convergence(n, e->v.rel_tol, e->v.abs_tol))) { vf_engine_destroy(n); return rc; }
    // (n works on e's stream already -- the caller's, or the handle's own, whose ownership moves to n with the swap below)
    if (e->v.x_max > 0 && e->x_used > 0) {
        // far between factors: the linear ones (they exist on the device only; their arrays do not depend on the capacity) copied,
        // the others re-sent from the host copies
        bool linear = false;
        for (int w = 0; w < e->v.B; w++) linear = linear || !e->h_lb[w].empty();
        if (linear) {
            if ((rc = n->ensure_far(e->x_used))) { vf_engine_destroy(n); return rc; }
            const size_t Bx = (size_t)B * VF_MAX_EXTRA;
            hipError_t he = hipMemcpyAsync(n->v.xl_n, e->v.xl_n, B * sizeof(int), hipMemcpyDeviceToDevice, n->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(n->v.xl_b, e->v.xl_b, Bx * sizeof(int), hipMemcpyDeviceToDevice, n->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(n->v.xl_U, e->v.xl_U, Bx * 6 * vf::XL_LD * sizeof(double), hipMemcpyDeviceToDevice, n->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(n->v.xl_r0, e->v.xl_r0, Bx * 6 * sizeof(double), hipMemcpyDeviceToDevice, n->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(n->v.xl_bx, e->v.xl_bx, Bx * 7 * sizeof(double), hipMemcpyDeviceToDevice, n->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(n->stream);
            if (he != hipSuccess) { vf_engine_destroy(n); return fail(VF_ERR_DEVICE, "vf_engine_grow: device copy failed: %s", hipGetErrorString(he)); }
            n->attach_far();
            n->h_lb = e->h_lb;
        }
        for (int w = 0; w < e->v.B; w++)
            if (e->h_xn[w] > 0 && (rc = vf_engine_set_extra_between(n, w, e->h_xn[w], e->h_xa[w].data(), e->h_xb[w].data(), e->h_xrec[w].data()))) {
                vf_engine_destroy(n);
                return rc;
            }
        n->recount_far();
    }
    // what the solver remembers from solve to solve: the non-monotone rule's damping / excursion state (a whole-history handle
    // grows again and again: each solve after a grow would otherwise restart from lambda0) and the far factors' counters
    if (e->v.x_best) {
        if ((rc = n->ensure_excursion())) { vf_engine_destroy(n); return rc; }
        hipError_t he = hipSuccess;
        auto cpx = [&](void* dst, const void* src, size_t bytes) { if (he == hipSuccess) he = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, n->stream); };
        cpx(n->v.carry, e->v.carry, B * sizeof(int));
        cpx(n->v.n_prov, e->v.n_prov, B * sizeof(int));
        cpx(n->v.prov, e->v.prov, B * sizeof(int));
        cpx(n->v.ref_cost, e->v.ref_cost, B * sizeof(double));
        for (int pl = 0; pl < 16 && he == hipSuccess; pl++)
            he = hipMemcpy2DAsync(n->v.x_best + (size_t)pl * G1, (size_t)M1 * sizeof(double), e->v.x_best + (size_t)pl * G0, (size_t)M0 * sizeof(double),
                                  (size_t)M0 * sizeof(double), (size_t)B, hipMemcpyDeviceToDevice, n->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(n->stream);
        if (he != hipSuccess) { vf_engine_destroy(n); return fail(VF_ERR_DEVICE, "vf_engine_grow: device copy failed: %s", hipGetErrorString(he)); }
    }
    n->far_transported = e->far_transported;
    n->far_ended = e->far_ended;
    n->far_absorbed = e->far_absorbed;
    n->marg_since_drop = e->marg_since_drop;
    n->inc_updates = e->inc_updates;
    n->inc_full = e->inc_full;
    n->own_stream = e->own_stream;
    std::swap(*e, *n);             // *e: the grown engine; *n: the old buffers
    n->own_stream = false;         // (the stream lives on in *e)
    // the asynchronous-staging resources (second stream, events, sticky words, pinned result block) move with the handle too
    std::swap(e->async_on, n->async_on);
    std::swap(e->stream2, n->stream2);
    std::swap(e->ev_fork, n->ev_fork);
    std::swap(e->ev_join, n->ev_join);
    std::swap(e->sticky_dev, n->sticky_dev);
    std::swap(e->res_host, n->res_host);
    vf_engine_destroy(n);
    cold(e);
    e->slid = e->redo = 0;
    e->epoch++;
    // the linearisation of the current states is part of the state other entry points rely on (vf_engine_marginalize reads
    // the Jacobians of the oldest keyframe's factors): recompute it in the new buffers
    bool any = false;
    for (int w = 0; w < e->v.B; w++) any = any || e->h_hi[w] > e->h_lo[w];
    if (any && (rc = vf_engine_linearize(e, 0))) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

int vf_engine_sync(vf_engine* e) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

// ------------------------------------------------------------------ read-back
static int read_sel(vf_engine* e, int window, int* sel) {
    HIPCHK(hipMemcpyAsync(sel, e->v.sel + window, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return VF_OK;
}

int vf_engine_read_imu_lin(vf_engine* e, int window, int which, int k0, int n, double* r15, double* J450) {
    DeviceGuard dev_guard_(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (n == 0) return VF_OK;
    int sel = 0;
    if ((rc = read_sel(e, window, &sel))) return rc;
    const int b = sel ^ (which ? 1 : 0);
    const size_t bytes = (size_t)n * vf::IMU_OUT * sizeof(double);
    if ((rc = e->ensure_stage(bytes))) return rc;
    vf::launch_gather_imu_lin(e->v, b, (long)window * e->v.M + k0, n, e->stage, e->stream);
    HIPCHK(hipGetLastError());
    std::vector<double> h((size_t)n * vf::IMU_OUT);
    HIPCHK(hipMemcpyAsync(h.data(), e->stage, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < n; i++) {
        if (r15) memcpy(r15 + (size_t)i * 15, h.data() + (size_t)i * vf::IMU_OUT, 15 * sizeof(double));
        if (J450) memcpy(J450 + (size_t)i * 450, h.data() + (size_t)i * vf::IMU_OUT + 15, 450 * sizeof(double));
    }
    return VF_OK;
}

int vf_engine_read_between_lin(vf_engine* e, int window, int which, int k0, int n, double* r6, double* Ja, double* Jb) {
    DeviceGuard dev_guard_(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (n == 0) return VF_OK;
    int sel = 0;
    if ((rc = read_sel(e, window, &sel))) return rc;
    const int b = sel ^ (which ? 1 : 0);
    const size_t bytes = (size_t)n * vf::BTW_OUT * sizeof(double);
    if ((rc = e->ensure_stage(bytes))) return rc;
    const double* src = e->v.btw_out + (size_t)b * (size_t)(e->v.G / 64) * vf::BTW_OUT * 64;
    vf::launch_gather(src, e->stage, (long)window * e->v.M + k0, n, vf::BTW_OUT, e->stream);
    HIPCHK(hipGetLastError());
    std::vector<double> h((size_t)n * vf::BTW_OUT);
    HIPCHK(hipMemcpyAsync(h.data(), e->stage, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < n; i++) {
        const double* p = h.data() + (size_t)i * vf::BTW_OUT;
        if (r6) memcpy(r6 + (size_t)i * 6, p, 6 * sizeof(double));
        if (Ja) memcpy(Ja + (size_t)i * 36, p + 6, 36 * sizeof(double));
        if (Jb) memcpy(Jb + (size_t)i * 36, p + 42, 36 * sizeof(double));
    }
    return VF_OK;
}

int vf_engine_read_normal(vf_engine* e, int window, int k0, int n, double* Hband, double* g15) {
    DeviceGuard dev_guard_(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (n == 0) return VF_OK;
    const size_t g0 = (size_t)window * e->v.M + k0;
    if (assembles_in_solve(e) || assembles_in_hybrid(e)) {
        // H and g are not kept by the solves of this engine: assemble them now, for this window only, whether or not the
        // termination rule has finished it (a converged window's H is what Engine.pose_information is asked for)
        vf::launch_assemble_window(e->v, window, e->stream);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (Hband) {
        // device rows are packed for the solver (vf_kernels.hpp "Block row of H") -> the documented [n][4][15][15]
        std::vector<double> raw((size_t)n * vf::HROW);
        HIPCHK(hipMemcpy(raw.data(), e->v.H + g0 * vf::HROW, raw.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::fill(Hband, Hband + (size_t)n * 900, 0.0);
        for (int k = 0; k < n; k++) {
            const double* r = raw.data() + (size_t)k * vf::HROW;
            double* o = Hband + (size_t)k * 900;
            for (int a = 0; a < 15; a++)
                for (int c = 0; c < 15; c++) {
                    o[a * 15 + c] = r[vf::H_D0 + (a >= c ? vf::h_tri(a, c) : vf::h_tri(c, a))];
                    o[225 + a * 15 + c] = r[vf::H_D1 + a * 15 + c];
                }
            for (int a = 0; a < 6; a++) {
                for (int c = 0; c < 6; c++) {
                    o[450 + a * 15 + c] = r[vf::H_D2 + a * 6 + c];
                    o[675 + a * 15 + c] = r[vf::H_D3 + a * 6 + c];
                }
                for (int c = 0; c < 9; c++) o[450 + a * 15 + 6 + c] = r[vf::H_DX + a * 9 + c];
            }
        }
    }
    if (g15) HIPCHK(hipMemcpy(g15, e->v.gvec + g0 * 15, (size_t)n * 15 * sizeof(double), hipMemcpyDeviceToHost));
    return VF_OK;
}

int vf_engine_read_delta(vf_engine* e, int window, int k0, int n, double* d) {
    DeviceGuard dev_guard_(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (n == 0 || !d) return VF_OK;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(d, e->v.delta + ((size_t)window * e->v.M + k0) * 15, (size_t)n * 15 * sizeof(double), hipMemcpyDeviceToHost));
    return VF_OK;
}

int vf_engine_read_panels(vf_engine* e, int window, int k0, int n, double* panels) {
    DeviceGuard dev_guard_(e);
    int rc = check_range(e, window, k0, n);
    if (rc) return rc;
    if (n == 0 || !panels) return VF_OK;
    HIPCHK(hipStreamSynchronize(e->stream));
    // packed device layout (vf_kernels.hpp "Cholesky panel") -> the documented [43][16] (column 15 = 0)
    std::vector<double> raw((size_t)n * vf::PANEL);
    HIPCHK(hipMemcpy(raw.data(), e->v.Lp + ((size_t)window * e->v.M + k0) * vf::PANEL, raw.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < n; k++)
        for (int r = 0; r < 43; r++)
            for (int c = 0; c < 16; c++) {
                double x = 0.0;
                if (c < 15) { const int i = vf::panel_idx(r, c); x = i == vf::PANEL_DUMP ? 0.0 : raw[(size_t)k * vf::PANEL + i]; }
                panels[(size_t)k * 688 + r * 16 + c] = x;
            }
    return VF_OK;
}

int vf_engine_read_lm(vf_engine* e, int window, double* cost, double* lambda, int* acc, int* rej, int* fails) {
    DeviceGuard dev_guard_(e);
    int rc = check_window(e, window);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (cost) HIPCHK(hipMemcpy(cost, e->v.cost + window, sizeof(double), hipMemcpyDeviceToHost));
    if (lambda) HIPCHK(hipMemcpy(lambda, e->v.lambda + window, sizeof(double), hipMemcpyDeviceToHost));
    if (acc) HIPCHK(hipMemcpy(acc, e->v.n_acc + window, sizeof(int), hipMemcpyDeviceToHost));
    if (rej) HIPCHK(hipMemcpy(rej, e->v.n_rej + window, sizeof(int), hipMemcpyDeviceToHost));
    if (fails) HIPCHK(hipMemcpy(fails, e->v.n_fail + window, sizeof(int), hipMemcpyDeviceToHost));
    return VF_OK;
}

// ------------------------------------------------------------------ measurement
int vf_engine_read_excursions(vf_engine* e, int window, int* provisional_trials, int* open_now) {
    DeviceGuard dev_guard_(e);
    int rc = check_window(e, window);
    if (rc) return rc;
    int np = 0, pr = 0;
    if (e->v.x_best) {
        HIPCHK(hipMemcpyAsync(&np, e->v.n_prov + window, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(&pr, e->v.prov + window, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    if (provisional_trials) *provisional_trials = np;
    if (open_now) *open_now = pr;
    return VF_OK;
}
int vf_engine_time_stage(vf_engine* e, int stage, int reps, float* avg_ms) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e || !avg_ms || reps < 1) return fail(VF_ERR_INVALID, "bad argument");
    vf::View tv = e->v;
    tv.stop_on = 0;           // stage timings are of the full work, whatever the windows' convergence flags say
    auto run = [&]() {
        switch (stage) {
            case VF_STAGE_LINEARIZE_IMU: vf::launch_linearize_imu(tv, 0, e->stream); break;
            case VF_STAGE_LINEARIZE_BTW: vf::launch_linearize_between(tv, 0, e->stream); break;
            case VF_STAGE_ASSEMBLE: vf::launch_assemble(tv, e->stream); break;
            case VF_STAGE_SOLVE: vf::launch_band_solve(tv, e->stream); break;
            case VF_STAGE_RETRACT: vf::launch_retract(tv, e->stream); break;
            case VF_STAGE_DECIDE: vf::launch_decide(tv, 1, e->stream); break;
            case VF_STAGE_ASSEMBLE_IDLE: vf::launch_assemble(tv, e->stream); break;
            default: break;
        }
    };
    if (stage < VF_STAGE_LINEARIZE_IMU || stage > VF_STAGE_ASSEMBLE_IDLE) return fail(VF_ERR_INVALID, "unknown stage %d", stage);
    // time the full-work form of K3 (inside iterate() it is skipped for windows whose last trial was rejected)
    if (stage == VF_STAGE_ASSEMBLE) HIPCHK(hipMemsetAsync(e->v.fresh, 0x01, e->v.B * sizeof(int), e->stream));
    if (stage == VF_STAGE_ASSEMBLE_IDLE) HIPCHK(hipMemsetAsync(e->v.fresh, 0, e->v.B * sizeof(int), e->stream));
    run();  // warm
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    for (int i = 0; i < reps; i++) run();
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    HIPCHK(hipEventSynchronize(e->ev1));
    HIPCHK(hipGetLastError());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    *avg_ms = ms / reps;
    if (stage == VF_STAGE_ASSEMBLE_IDLE) HIPCHK(hipMemsetAsync(e->v.fresh, 0x01, e->v.B * sizeof(int), e->stream));   // H is as it was
    return VF_OK;
}

int vf_engine_time_iterate(vf_engine* e, int iterations, float* ms) {
    DeviceGuard dev_guard_(e);
    if (e) cold(e);
    if (!e || !ms) return fail(VF_ERR_INVALID, "bad argument");
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    int rc = vf_engine_iterate(e, iterations);
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    HIPCHK(hipEventSynchronize(e->ev1));
    HIPCHK(hipEventElapsedTime(ms, e->ev0, e->ev1));
    return VF_OK;
}

int vf_engine_counts(vf_engine* e, int64_t* n_imu, int64_t* n_btw, int64_t* n_kf) {
    DeviceGuard dev_guard_(e);
    if (!e) return fail(VF_ERR_INVALID, "engine is null");
    HIPCHK(hipStreamSynchronize(e->stream));
    std::vector<int> a((size_t)e->v.G);
    HIPCHK(hipMemcpy(a.data(), e->v.btw_a, a.size() * sizeof(int), hipMemcpyDeviceToHost));
    int64_t ni = 0, nb = 0, nk = 0;
    for (int w = 0; w < e->v.B; w++) {
        const int lo = e->h_lo[w], hi = e->h_hi[w];
        if (hi <= lo) continue;
        nk += hi - lo;
        ni += hi - lo - 1;
        for (int k = lo + 1; k < hi; k++) {
            const int aa = a[(size_t)w * e->v.M + k];
            if (aa >= lo && aa < k) nb++;
        }
    }
    if (n_imu) *n_imu = ni;
    if (n_btw) *n_btw = nb;
    if (n_kf) *n_kf = nk;
    return VF_OK;
}

}  // extern "C"
