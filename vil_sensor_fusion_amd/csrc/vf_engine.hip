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
    // The adaptive trial loop of vf_engine_iterate reads the whole result block where it used to read the `done` flag: when the flag
    // is up the solve is over and the block IS the result -- the vf_engine_read_result that follows finds it here (res_cached, until
    // any other entry point runs) and costs no second synchronisation.  Sticky words an early read has consumed wait in res_carry.
    bool res_cached = false;
    int res_slot = -1, res_carry[2] = {0, 0};
    int ensure_async() {          // the sticky words and the pinned result block (every engine may be asked for a vf_engine_read_result)
        if (res_host) return VF_OK;
        HIPCHK(hipMalloc((void**)&sticky_dev, 2 * sizeof(int)));
        HIPCHK(hipMemsetAsync(sticky_dev, 0, 2 * sizeof(int), stream));
        HIPCHK(hipHostMalloc((void**)&res_host, sizeof(vf::SolveResult), hipHostMallocDefault));
        return VF_OK;
    }
    int ensure_side() {           // the second stream (hipStreamCreate costs 9 ms: only engines that stage asynchronously get one)
        if (stream2) return VF_OK;
        HIPCHK(hipStreamCreate(&stream2));
        HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        return VF_OK;
    }
    int join_side() {
        if (!side_open) return VF_OK;
        side_open = false;
        HIPCHK(hipEventRecord(ev_join, stream2));
        HIPCHK(hipStreamWaitEvent(stream, ev_join, 0));
        return VF_OK;
    }
    // async_base: staging calls enqueue and return.  async_now: ... and the window holds no far factors -- the marginalisation may run
    // on the second stream / ahead of time (k_marginalize<0> only; with far factors alive it reads and rewrites their lists, which
    // the host re-sends on the main stream).  async_far: the rest -- the marginalisation enqueued on the main stream.
    bool async_base() const { return async_on && v.B == 1 && own_stream; }
    bool async_now() const { return async_base() && x_used == 0; }
    bool async_far() const { return async_base() && x_used > 0; }
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
    std::vector<char> h_xdirty;                        // the device's copy of a window's list may differ from the host's (k_marginalize has
                                                       // struck out entries the host has not yet): vf_engine_set_extra_between must re-send
    std::vector<std::vector<double>> h_xrec;
    double* x_gtmp = nullptr;
    double* x_Z = nullptr;
    size_t x_zstride = 0;
    long far_transported = 0, far_ended = 0, far_absorbed = 0;   // far factors moved on to the next keyframe when theirs left the window /
                                                                 // dropped without a marginalisation / absorbed into the marginal prior
    bool marg_since_drop = false;   // vf_engine_marginalize ran since the last vf_engine_drop_oldest (the GraphManager's call pair)
    int x_zslots = 0;          // slots x_Z holds columns for (6 columns each); grown on demand, never beyond x_cap
    int x_cap = VF_MAX_EXTRA;  // far factors a window may hold (vf_engine_opts.max_far_factors): the width of the device lists
    // the device lists are allocated once (they stay in `allocs`); a failure half way leaves what exists in place and a
    // later call picks up from there (no second allocation, nothing leaked)
    int *x_la = nullptr, *x_lb = nullptr;
    double *x_li = nullptr, *x_lo = nullptr;
    // Woodbury columns of a single-window engine as one batch: a second engine of up to 6 x x_cap windows and the same
    // capacity, on this engine's stream; window q holds a copy of the window's H and column q of U as right-hand side
    // (k_cols_prepare), one partitioned solve of it returns all of Z.  Made when the first far factor arrives, never inside a
    // solve (a solve may be under stream capture); absent (sequential columns, as on batch engines) if it cannot be had.
    vf_engine* far_columns = nullptr;
    bool is_far_columns = false;
    bool far_columns_off = false;   // a column engine could not be made: the columns are solved one after the other
    // linear far factors (View::xl_*; made and kept by k_marginalize): the host mirrors only their number and far ends
    std::vector<std::vector<int>> h_lb;
    int h_ln(int w) const { return h_lb.empty() ? 0 : (int)h_lb[w].size(); }
    void attach_far() {              // after ensure_far: the View sees the slot arrays, the host mirrors exist
        if (v.x_max) return;
        v.x_a = x_la; v.x_b = x_lb; v.x_in = x_li; v.x_out = x_lo;
        v.x_max = x_cap;
        v.far_big = tune.far_big_forms && x_cap > vf::MAX_EXTRA ? 1 : 0;
        h_xn.assign(v.B, 0);
        h_xdirty.assign(v.B, 1);
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
        const size_t B = (size_t)v.B, X = (size_t)x_cap;
        int rc;
        if (!x_la) { if ((rc = alloc(&x_la, B * X, false))) return rc; HIPCHK(hipMemsetAsync(x_la, 0xff, B * X * sizeof(int), stream)); }
        if (!x_lb) { if ((rc = alloc(&x_lb, B * X, false))) return rc; HIPCHK(hipMemsetAsync(x_lb, 0xff, B * X * sizeof(int), stream)); }
        if (!x_li && (rc = alloc(&x_li, B * X * vf::BTW_IN))) return rc;
        if (!x_lo && (rc = alloc(&x_lo, 2 * B * X * vf::BTW_OUT))) return rc;
        if ((!v.xl_n && (rc = alloc(&v.xl_n, B))) || (!v.xl_b && (rc = alloc(&v.xl_b, B * X))) || (!v.xl_U && (rc = alloc(&v.xl_U, B * X * 6 * (27 + 6 * X)))) ||
            (!v.xl_r0 && (rc = alloc(&v.xl_r0, B * X * 6))) || (!v.xl_bx && (rc = alloc(&v.xl_bx, B * X * 7))) || (!v.xl_out && (rc = alloc(&v.xl_out, 2 * B * X * 6)))) return rc;
        // (more far factors than the LDS forms of k_marginalize / k_extra_combine hold: their systems live here)
        // (the others: room for the solution of the Woodbury system, handed from k_extra_combine to k_extra_apply)
        if (!v.far_scratch && (rc = alloc(&v.far_scratch, B * (x_cap > vf::MAX_EXTRA ? vf::FAR_SCRATCH : (size_t)64)))) return rc;
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
    const_cast<vf_engine*>(e)->res_cached = false;
    want = e->opts.device;
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipSetDevice(want); return; }
    if (prev != want) (void)hipSetDevice(want);
}

extern "C" {
#include "engine/engine_create.inc"
#include "engine/engine_staging.inc"
#include "engine/engine_solve.inc"
#include "engine/engine_shard.inc"
#include "engine/engine_misc.inc"
#include "engine/engine_compat.inc"
#include "engine/engine_window.inc"
#include "engine/engine_read.inc"
}  // extern "C"
