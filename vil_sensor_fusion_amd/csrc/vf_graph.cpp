// vf_graph.cpp -- host-side mirror of VILFusion::GraphManager + VILFusion::IMUManager behind the
// C ABI (include/vilfusion.h "GraphManager surface").
//
// Only bookkeeping lives here: key numbering, the IMU sample deque and how IMUManager::getFactor
// cuts it (which samples are dropped / integrated / interpolated), the queue of not-yet-added
// IMU factors, staged between factors, the two mutexes and the callbacks.  Every number that
// depends on the factor math -- preintegration (K0), initial-value prediction, linearisation,
// the LM solve -- is computed by the HIP engine; there is no CPU fallback.
//
// Reference: gtsam_fusion/src/gtsam_fusion/GraphManager.cpp, IMUManager.cpp.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vilfusion.h"

namespace {

struct ImuSample { double t, acc[3], gyro[3]; };
struct PendingImu {            // one queued CombinedImuFactor (GraphManager::_imuQueue)
    uint64_t key;              // X(key-1) -> X(key)
    std::vector<double> steps; // 7 per step: dt, acc, gyro
    double bias[6];            // getBias() at reserveNode time (GraphManager.cpp:61)
    std::vector<double> record; // non-empty: a ready-made factor handed in through vf_add_imu_factor (addFactor), 190 doubles
    bool staged = false;        // preintegrated on the device already, at reserveNode time (where the reference preintegrates: GraphManager.cpp:59-66)
};
struct PendingBetween { uint64_t a, b; double rec[VF_BTW_RECORD]; bool on_device = false; };
// a between factor as the caller handed it in: what GraphManager::graph() shows until the next solve (vf_graph_get_staged)
struct StagedFactor { uint64_t a, b; double q[4], t[3], cov[36]; };

// R upper-triangular, R^T R = cov^-1 (noiseModel::Gaussian::Covariance, SensorManagerRos.cpp:99).
// 6x6 noise-model construction is factor *construction*, done once per measurement on the host
// exactly where the reference does it.
bool sqrt_info_upper6(const double* cov, double* Rp) {
    // reverse Cholesky cov = U U^T (U upper) then R = U^-1
    double U[36] = {0}, T[36] = {0};
    for (int j = 5; j >= 0; j--) {
        double d = cov[j * 6 + j];
        for (int l = j + 1; l < 6; l++) d -= U[j * 6 + l] * U[j * 6 + l];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        const double ujj = std::sqrt(d);
        U[j * 6 + j] = ujj;
        for (int i = 0; i < j; i++) {
            double a = 0.5 * (cov[i * 6 + j] + cov[j * 6 + i]);
            for (int l = j + 1; l < 6; l++) a -= U[i * 6 + l] * U[j * 6 + l];
            U[i * 6 + j] = a / ujj;
        }
    }
    for (int c = 0; c < 6; c++) {
        T[c * 6 + c] = 1.0 / U[c * 6 + c];
        for (int r = c - 1; r >= 0; r--) {
            double a = 0.0;
            for (int l = r + 1; l <= c; l++) a += U[r * 6 + l] * T[l * 6 + c];
            T[r * 6 + c] = -a / U[r * 6 + r];
        }
    }
    int o = 0;
    for (int r = 0; r < 6; r++)
        for (int c = r; c < 6; c++) Rp[o++] = T[r * 6 + c];
    return true;
}

}  // namespace

struct vf_graph {
    vf_engine* eng = nullptr;
    vf_graph_opts opts{};
    vf_imu_params imu{};
    // guarded by graph_mutex (GraphManager::_graphMutex + IMUManager::_bufferMutex)
    std::mutex graph_mutex, buffer_mutex;
    std::mutex solve_mutex;        // one vf_solve at a time (order: solve -> graph -> state; nothing else takes it)
    std::deque<ImuSample> buffer;
    std::deque<PendingImu> imu_queue;
    std::vector<PendingBetween> staged_between;
    // Between factors the band cannot hold -- wider than VF_MAX_BANDWIDTH keyframes, or a second one ending at a key (loop
    // closures; iSAM2 takes any pair of keys, GraphManager.cpp:83-88): handed to the engine as "far" factors
    // (vf_engine_set_extra_between) at every solve.  When the older key of one leaves the fixed-lag window the engine
    // marginalises the factor with it and keeps it from then on as linear rows of its own (include/vilfusion.h): the entry
    // here goes (the list is replaced by what vf_engine_get_extra_between reports).  band_end[k] != 0: key k
    // already carries a band factor.  far_new counts the ones added since the last solve (they are part of graph()->size()).
    std::vector<PendingBetween> far_between;
    std::deque<uint8_t> band_end;      // entry i: key band_base + i (trimmed below the window at every solve)
    uint64_t band_base = 0;
    bool has_band_end(uint64_t k) const { return k >= band_base && k - band_base < band_end.size() && band_end[k - band_base]; }
    void set_band_end(uint64_t k, uint8_t v) {
        if (k < band_base) return;
        if (k - band_base >= band_end.size()) { if (!v) return; band_end.resize(k - band_base + 1, 0); }
        band_end[k - band_base] = v;
    }
    int far_new = 0;
    bool far_on_device = false;    // the engine holds a non-empty far list (written under solve_mutex only)
    std::atomic<int> far_linear{0};  // far ends of the engine's linear far factor (far factors marginalised with their older key): they share opts.max_far_factors
    int staged_count = 3;  // the three priors (GraphManager.cpp:33-35)
    bool priors_staged = true;              // ... which the first solve takes with everything else (_graph->resize(0), :114)
    std::vector<StagedFactor> staged_log;   // the between factors among them, in the order addBetweenFactor took them
    uint64_t current_key = 0;
    double last_pose_time = -1.0;
    std::vector<double> key_time;  // time of each reserved key
    // guarded by state_mutex (GraphManager::_stateMutex)
    std::mutex state_mutex;
    double state[16] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double anchor[16] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // X(0), V(0), B(0) and the means of their priors
    uint64_t solved_key = 0;  // keys [0, solved_key] hold states on the device
    int lo = 0;               // slot of the oldest keyframe in the window
    uint64_t key_base = 0;    // slot = key - key_base (advances by multiples of 64 on compaction)
    std::vector<std::pair<vf_callback, void*>> callbacks;
};

extern "C" {

// defined in vf_engine.hip: graph-level messages go to the same thread-local vf_last_error()
void vf_set_last_error_(const char* msg);

static int gerr(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    vf_set_last_error_(buf);
    return code;
}

#undef vf_graph_default_opts
#undef vf_engine_default_opts
static void graph_defaults(vf_graph_opts* o) {
    memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t)sizeof(*o);
    o->capacity = 4096;
    o->lag = 0;
    o->iterations = 5;
    o->device = 0;
    // GraphManager.cpp:27-31: rad,rad,rad,m,m,m ; m/s ; bias
    const double s[15] = {1e-6, 1e-6, 1e-6, 5e-5, 5e-5, 5e-5, 1e-5, 1e-5, 1e-5, 1e-7, 1e-7, 1e-7, 1e-7, 1e-7, 1e-7};
    memcpy(o->prior_sigma, s, sizeof(s));
    o->rel_tol = 1e-5;   // gtsam::LevenbergMarquardtParams relativeErrorTol / absoluteErrorTol
    o->abs_tol = 1e-5;
    o->cold_start = 0;
    o->fixed_capacity = 0;
    o->reference_compat = 0;
    o->relin_threshold = 1e-4;   // GraphManager.cpp:40
    o->incremental = 0;
    o->wildfire = 0.0;
    o->min_model_fidelity = 0.0;
    o->synchronous_staging = 0;
    o->max_far_factors = 0;      // = VF_MAX_FAR_LIMIT
}
// (the caller's struct may be shorter than the library's: include/vilfusion.h "struct_size")
void vf_graph_default_opts_sized(vf_graph_opts* o, uint32_t struct_size) {
    if (!o || struct_size < sizeof(uint32_t)) return;
    vf_graph_opts full;
    graph_defaults(&full);
    const uint32_t n = struct_size < sizeof(full) ? struct_size : (uint32_t)sizeof(full);
    memcpy(o, &full, n);
    o->struct_size = n;
}
void vf_graph_default_opts(vf_graph_opts* o) { vf_graph_default_opts_sized(o, (uint32_t)sizeof(vf_graph_opts)); }

int vf_create(const vf_imu_params* imu, const vf_graph_opts* opts, vf_graph** out) {
    if (!imu || !out) return gerr(VF_ERR_INVALID, "null argument");
    vf_graph_opts o;
    graph_defaults(&o);
    if (opts) {
        if (opts->struct_size < 8 || opts->struct_size > sizeof(o))
            return gerr(VF_ERR_INVALID, "vf_graph_opts.struct_size = %u: this library knows sizes up to %zu (fill the struct with vf_graph_default_opts; "
                        "a caller built against a newer header needs a newer library)", opts->struct_size, sizeof(o));
        memcpy(&o, opts, opts->struct_size);
        o.struct_size = (uint32_t)sizeof(o);
    }
    if (o.capacity < 8 || o.iterations < 0 || o.lag < 0) return gerr(VF_ERR_INVALID, "bad graph options");
    if (o.lag != 0 && o.lag < 4) return gerr(VF_ERR_INVALID, "lag must be 0 or >= 4 keyframes");
    if (!(o.rel_tol >= 0.0) || !(o.abs_tol >= 0.0)) return gerr(VF_ERR_INVALID, "tolerances must be >= 0");
    if (o.reference_compat && o.lag != 0) return gerr(VF_ERR_INVALID, "reference_compat needs lag == 0 (the reference's graph is unbounded)");
    if (o.reference_compat && !(o.relin_threshold >= 0.0)) return gerr(VF_ERR_INVALID, "relin_threshold must be >= 0");
    if (o.incremental && !o.reference_compat) return gerr(VF_ERR_INVALID, "incremental needs reference_compat (the update it makes incremental is the iSAM2-like one)");
    if (o.incremental && !(o.wildfire >= 0.0)) return gerr(VF_ERR_INVALID, "wildfire must be >= 0");
    if (o.max_far_factors < 0 || o.max_far_factors > VF_MAX_FAR_LIMIT) return gerr(VF_ERR_INVALID, "max_far_factors must be in 0..%d", VF_MAX_FAR_LIMIT);
    // (a handle takes what the library can hold: while eight or fewer are alive the solver's forms are those of an engine made for
    // eight, whatever the capacity; the lists are allocated when the first far factor arrives)
    if (o.max_far_factors == 0) o.max_far_factors = VF_MAX_FAR_LIMIT;
    const double covs[6] = {imu->acc_cov, imu->gyro_cov, imu->integration_cov, imu->bias_acc_cov, imu->bias_omega_cov, imu->bias_acc_omega_int};
    for (double c : covs)
        if (!(c > 0.0) || !std::isfinite(c)) return gerr(VF_ERR_NOT_SPD, "IMU covariances must be finite and > 0");
    // the engine allocates keyframe slots in whole AoSoA tiles of 64: keep the number the growth / compaction / capacity
    // checks below work with equal to what the engine really has (a handle created with capacity 16 has 64 slots; growing
    // it "to 32" would ask vf_engine_grow for the 64 it already has)
    o.capacity = (o.capacity + 63) / 64 * 64;
    vf_engine_opts eo;
    vf_engine_default_opts(&eo);
    eo.windows = 1;
    eo.capacity = o.capacity;
    eo.device = o.device;
    eo.cold_start = o.cold_start;
    eo.min_model_fidelity = o.min_model_fidelity > 0.0 ? o.min_model_fidelity : 0.0;
    eo.max_far_factors = o.max_far_factors;
    if (o.incremental) {
        // (the refined solve corrects a whole-window factorisation through J; the incremental update keeps the panels of an
        // elimination that ran forward in time from the anchor prior -- every pivot block is the conditional information of a
        // keyframe given its past, well conditioned whatever the history's length -- and does without)
        eo.incremental = o.incremental == 2 ? 2 : 1;
        eo.wildfire = o.wildfire;
        eo.refine_iterations = 0;
        eo.lm_excursion = 0;
    }
    vf_engine* eng = nullptr;
    int rc = vf_engine_create(&eo, &eng);
    if (rc) return rc;
    vf_graph* g = new vf_graph();
    g->eng = eng;
    g->opts = o;
    g->imu = *imu;
    g->key_time.push_back(-1.0);
    // X(0)=identity, V(0)=0, B(0)=0 with the three priors (GraphManager.cpp:20-35)
    double rec[VF_PRIOR_RECORD] = {0};
    rec[0] = 1.0;
    memcpy(rec + 16, o.prior_sigma, sizeof(double) * 15);
    if ((rc = vf_engine_set_states(eng, 0, 0, 1, g->state)) || (rc = vf_engine_set_prior(eng, 0, 0, rec)) ||
        (rc = vf_engine_set_range(eng, 0, 0, 1)) || (rc = vf_engine_set_convergence(eng, o.rel_tol, o.abs_tol)) ||
        (rc = vf_engine_set_async(eng, o.synchronous_staging ? 0 : 1))) {
        vf_engine_destroy(eng);
        delete g;
        return rc;
    }
    *out = g;
    return VF_OK;
}

void vf_destroy(vf_graph* g) {
    if (!g) return;
    vf_engine_destroy(g->eng);
    delete g;
}

int vf_set_initial_state(vf_graph* g, const double state16[16]) {
    if (!g || !state16) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->graph_mutex);
    std::lock_guard<std::mutex> sl(g->state_mutex);
    if (g->current_key != 0 || g->solved_key != 0) return gerr(VF_ERR_INVALID, "the initial state can only be set before the first node is reserved");
    double n = 0.0;
    for (int i = 0; i < 4; i++) n += state16[i] * state16[i];
    for (int i = 0; i < 16; i++)
        if (!std::isfinite(state16[i])) return gerr(VF_ERR_INVALID, "initial state entry %d is not finite", i);
    if (!(n > 0.0)) return gerr(VF_ERR_INVALID, "initial rotation is not a quaternion");
    double st[16];
    memcpy(st, state16, sizeof(st));
    for (int i = 0; i < 4; i++) st[i] /= std::sqrt(n);
    double rec[VF_PRIOR_RECORD];
    memcpy(rec, st, sizeof(double) * 16);
    memcpy(rec + 16, g->opts.prior_sigma, sizeof(double) * 15);
    int rc;
    if ((rc = vf_engine_set_states(g->eng, 0, 0, 1, st)) || (rc = vf_engine_set_prior(g->eng, 0, 0, rec))) return rc;
    memcpy(g->state, st, sizeof(st));
    memcpy(g->anchor, st, sizeof(st));
    return VF_OK;
}

int vf_add_imu(vf_graph* g, double time, const double acc[3], const double gyro[3]) {
    if (!g || !acc || !gyro) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->buffer_mutex);  // IMUManager.cpp:21
    ImuSample s;
    s.t = time;
    memcpy(s.acc, acc, sizeof(s.acc));
    memcpy(s.gyro, gyro, sizeof(s.gyro));
    g->buffer.push_back(s);
    return VF_OK;
}

// IMUManager::getFactor (IMUManager.cpp:27-74): which samples are dropped, integrated and
// interpolated.  The integration itself happens on the device at solve time (K0).
static void cut_imu_segment(vf_graph* g, double start, double end, std::vector<double>& steps) {
    std::lock_guard<std::mutex> lk(g->buffer_mutex);
    ImuSample prev{};
    while (!g->buffer.empty() && g->buffer.front().t <= start) {  // :35-40
        prev = g->buffer.front();
        g->buffer.pop_front();
    }
    prev.t = start;  // :44
    while (!g->buffer.empty() && g->buffer.front().t < end) {  // :46-54
        const ImuSample m = g->buffer.front();
        g->buffer.pop_front();
        const double st[7] = {m.t - prev.t, m.acc[0], m.acc[1], m.acc[2], m.gyro[0], m.gyro[1], m.gyro[2]};
        steps.insert(steps.end(), st, st + 7);
        prev = m;
    }
    if (!g->buffer.empty()) {  // :57-66, the interpolated sample stays in the buffer
        const ImuSample& f = g->buffer.front();
        const double w = (end - prev.t) / (f.t - prev.t);
        double st[7] = {end - prev.t, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            st[1 + i] = w * f.acc[i] + (1.0 - w) * prev.acc[i];
            st[4 + i] = w * f.gyro[i] + (1.0 - w) * prev.gyro[i];
        }
        steps.insert(steps.end(), st, st + 7);
    }
}

int vf_reserve_node(vf_graph* g, double time, uint64_t* key_out) {
    if (!g || !key_out) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->graph_mutex);  // GraphManager.cpp:54
    if (g->opts.lag == 0 && g->opts.fixed_capacity && (int)(g->current_key + 1) >= g->opts.capacity)
        return gerr(VF_ERR_CAPACITY, "keyframe capacity %d exhausted (fixed_capacity; set a lag to smooth indefinitely)", g->opts.capacity);
    double start;
    if (g->current_key + 1 > 1) {
        start = g->last_pose_time;  // :59-61
    } else {
        // first node integrates from the oldest buffered sample (IMUManager.cpp:76-79)
        std::lock_guard<std::mutex> bl(g->buffer_mutex);
        if (g->buffer.empty()) return gerr(VF_ERR_INDETERMINATE, "reserveNode before any IMU measurement");
        start = g->buffer.front().t;
    }
    PendingImu p;
    p.key = g->current_key + 1;
    {
        std::lock_guard<std::mutex> sl(g->state_mutex);  // getBias(): lock order graph -> state
        memcpy(p.bias, g->state + 10, sizeof(p.bias));
    }
    cut_imu_segment(g, start, time, p.steps);
    // a factor without a single IMU step has no covariance: GTSAM would fail at the solve with the factor already in
    // the graph; here the node is refused and nothing is queued (checked before the key is consumed)
    if (p.steps.empty()) return gerr(VF_ERR_INDETERMINATE, "reserveNode(%.6f): no IMU measurement in (%.6f, %.6f]", time, start, time);
    {
        // The reference preintegrates here, in reserveNode (IMUManager::getFactor, GraphManager.cpp:59-66); so does this: K0 of the
        // new factor is enqueued on the engine's stream now, asynchronously (vf_engine_set_async), and the solve finds the
        // record in place instead of starting with a 60 us kernel.  Only when the slot exists already (no growth / compaction
        // pending) and every factor queued in front of it is on the device too (slots are consecutive); _stateMutex orders this
        // against a solve in flight, as getBias() above does.
        std::lock_guard<std::mutex> sl(g->state_mutex);
        const long slot = (long)(p.key - g->key_base);
        bool prev_ok = true;
        for (const auto& q : g->imu_queue) prev_ok = prev_ok && q.staged;
        if (prev_ok && !g->opts.synchronous_staging && slot >= 1 && slot < g->opts.capacity && p.key > g->key_base) {
            const int32_t off[2] = {0, (int32_t)(p.steps.size() / 7)};
            if (vf_engine_preintegrate(g->eng, 0, (int)slot, 1, off, p.steps.data(), p.bias, &g->imu) == VF_OK) p.staged = true;
        }
    }
    g->current_key++;
    g->imu_queue.push_back(std::move(p));
    g->last_pose_time = time;
    g->key_time.push_back(time);
    *key_out = g->current_key;
    return VF_OK;
}

int vf_add_imu_factor(vf_graph* g, uint64_t key, const double* rec190) {
    if (!g || !rec190) return gerr(VF_ERR_INVALID, "null argument");
    if (!(rec190[0] > 0.0) || !std::isfinite(rec190[0])) return gerr(VF_ERR_INVALID, "imu factor with deltaTij = %g", rec190[0]);
    for (int i = 0; i < VF_IMU_RECORD; i++)
        if (!std::isfinite(rec190[i])) return gerr(VF_ERR_INVALID, "imu factor record entry %d is not finite", i);
    for (int r = 0, o = 70; r < 15; o += 15 - r, r++)          // diagonal of the packed upper-triangular square-root information
        if (!(rec190[o] > 0.0)) return gerr(VF_ERR_NOT_SPD, "imu factor: square-root information has a non-positive diagonal entry (row %d)", r);
    std::lock_guard<std::mutex> lk(g->graph_mutex);  // GraphManager.cpp:92
    if (key != g->current_key + 1)
        return gerr(VF_ERR_BAD_KEY, "imu factor must end at the next key %llu (got %llu): keys are consecutive", (unsigned long long)(g->current_key + 1), (unsigned long long)key);
    if (g->opts.lag == 0 && g->opts.fixed_capacity && (int)(g->current_key + 1) >= g->opts.capacity) return gerr(VF_ERR_CAPACITY, "keyframe capacity %d exhausted", g->opts.capacity);
    PendingImu p;
    p.key = key;
    memset(p.bias, 0, sizeof(p.bias));
    p.record.assign(rec190, rec190 + VF_IMU_RECORD);
    g->current_key++;
    g->imu_queue.push_back(std::move(p));
    g->last_pose_time = (g->last_pose_time < 0.0 ? 0.0 : g->last_pose_time) + rec190[0];
    g->key_time.push_back(g->last_pose_time);
    return VF_OK;
}

int vf_get_most_recent_estimate(vf_graph* g, double q[4], double t[3], double v[3]) {
    if (!g) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->graph_mutex);  // GraphManager.cpp:79
    // The reference never assigns _mostRecentEstimate (GraphManager.h:109; no write in GraphManager.cpp): the member
    // stays the default NavState -- identity pose, zero velocity -- for the life of the node.  Mirrored as it is.
    if (q) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; }
    if (t) t[0] = t[1] = t[2] = 0.0;
    if (v) v[0] = v[1] = v[2] = 0.0;
    return VF_OK;
}

int vf_most_recent_pose_time(vf_graph* g, double* time, uint64_t* key) {
    if (!g) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->graph_mutex);
    if (time) *time = g->last_pose_time;
    if (key) *key = g->current_key;
    return VF_OK;
}

int vf_add_between(vf_graph* g, uint64_t prev, uint64_t cur, const double q[4], const double t[3], const double cov[36]) {
    if (!g || !q || !t || !cov) return gerr(VF_ERR_INVALID, "null argument");
    PendingBetween b;
    b.a = prev;
    b.b = cur;
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (!(n > 0.0) || !std::isfinite(n)) return gerr(VF_ERR_INVALID, "between rotation is not a quaternion");
    for (int i = 0; i < 4; i++) b.rec[i] = q[i] / n;  // gtsam::Rot3(w,x,y,z) normalises
    for (int i = 0; i < 3; i++) b.rec[4 + i] = t[i];
    if (!sqrt_info_upper6(cov, b.rec + 7)) return gerr(VF_ERR_NOT_SPD, "between covariance is not symmetric positive definite");
    std::lock_guard<std::mutex> lk(g->graph_mutex);  // GraphManager.cpp:85
    if (prev >= cur || cur > g->current_key) return gerr(VF_ERR_BAD_KEY, "between factor keys (%llu, %llu) not reserved in order", (unsigned long long)prev, (unsigned long long)cur);
    bool band = cur - prev <= VF_MAX_BANDWIDTH && !g->has_band_end(cur);
    for (const auto& s : g->staged_between)
        if (s.b == cur) band = false;
    if (band) {
        g->staged_between.push_back(b);
    } else {
        // not a band factor: a far factor, solved as a low-rank correction (slower; include/vilfusion.h).  The list is bounded.
        if ((int)g->far_between.size() + g->far_linear.load() >= g->opts.max_far_factors)
            return gerr(VF_ERR_CAPACITY, "between factor (%llu, %llu) spans %llu keyframes or shares its end key, and the window already holds %d such factors "
                        "(vf_graph_opts.max_far_factors; at most %d)", (unsigned long long)prev, (unsigned long long)cur, (unsigned long long)(cur - prev),
                        g->opts.max_far_factors, VF_MAX_FAR_LIMIT);
        g->far_between.push_back(b);
        g->far_new++;
    }
    g->staged_count++;
    StagedFactor sf;
    sf.a = prev;
    sf.b = cur;
    memcpy(sf.q, b.rec, sizeof(sf.q));
    memcpy(sf.t, t, sizeof(sf.t));
    memcpy(sf.cov, cov, sizeof(sf.cov));
    g->staged_log.push_back(sf);
    return VF_OK;
}

int vf_set_callback(vf_graph* g, vf_callback cb, void* user) {
    if (!g || !cb) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->state_mutex);
    g->callbacks.emplace_back(cb, user);
    return VF_OK;
}

int vf_graph_staged(vf_graph* g, int* staged, int* queued) {
    if (!g) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->graph_mutex);
    if (staged) *staged = g->staged_count;
    if (queued) *queued = (int)g->imu_queue.size();
    return VF_OK;
}

// GraphManager::graph() (GraphManager.cpp:46-49): the factors staged since the last solve, by index -- at first the three priors
// of :27-35 (kinds 0, 1, 2 on key 0), then the between factors in the order they were added (kind 3).
int vf_graph_get_staged(vf_graph* g, int index, int* kind, uint64_t* key1, uint64_t* key2, double q[4], double t[3], double cov36[36]) {
    if (!g) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->graph_mutex);
    const int np = g->priors_staged ? 3 : 0;
    if (index < 0 || index >= np + (int)g->staged_log.size()) return gerr(VF_ERR_BAD_KEY, "staged factor %d of %d", index, np + (int)g->staged_log.size());
    if (q) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; }
    if (t) t[0] = t[1] = t[2] = 0.0;
    if (cov36) memset(cov36, 0, sizeof(double) * 36);
    if (index < np) {
        // the means are the anchor state (identity / zero, or vf_set_initial_state's); sigmas vf_graph_opts.prior_sigma
        std::lock_guard<std::mutex> sl(g->state_mutex);
        const double* s = g->opts.prior_sigma;
        if (kind) *kind = index;
        if (key1) *key1 = 0;
        if (key2) *key2 = 0;
        if (index == 0) {
            if (q) memcpy(q, g->anchor, sizeof(double) * 4);
            if (t) memcpy(t, g->anchor + 4, sizeof(double) * 3);
            if (cov36) for (int i = 0; i < 6; i++) cov36[i * 6 + i] = s[i] * s[i];
        } else if (index == 1) {
            if (t) memcpy(t, g->anchor + 7, sizeof(double) * 3);
            if (cov36) for (int i = 0; i < 3; i++) cov36[i * 6 + i] = s[6 + i] * s[6 + i];
        } else {
            // (a bias has six components: acc in t, gyro in q[1..3]; documented in vilfusion.h)
            if (t) memcpy(t, g->anchor + 10, sizeof(double) * 3);
            if (q) { q[0] = 0.0; memcpy(q + 1, g->anchor + 13, sizeof(double) * 3); }
            if (cov36) for (int i = 0; i < 6; i++) cov36[i * 6 + i] = s[9 + i] * s[9 + i];
        }
        return VF_OK;
    }
    const StagedFactor& f = g->staged_log[(size_t)(index - np)];
    if (kind) *kind = 3;
    if (key1) *key1 = f.a;
    if (key2) *key2 = f.b;
    if (q) memcpy(q, f.q, sizeof(f.q));
    if (t) memcpy(t, f.t, sizeof(f.t));
    if (cov36) memcpy(cov36, f.cov, sizeof(f.cov));
    return VF_OK;
}

int vf_solve(vf_graph* g) {
    if (!g) return gerr(VF_ERR_INVALID, "null argument");
    static const bool timing = getenv("VF_SOLVE_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_a = now();
    auto lap = [&](const char* what) { if (timing) { auto t_b = now(); fprintf(stderr, "[vf_solve] %-12s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t_b - t_a).count()); t_a = t_b; } };
    // ---- under _graphMutex: emptyImuQueue + snapshot of the staged factors (GraphManager.cpp:104-114)
    std::deque<PendingImu> imus;
    std::vector<PendingBetween> betweens, fars;
    uint64_t last_key;
    double last_time;
    int staged_before, far_new_before = 0;
    std::vector<StagedFactor> log_before;
    bool priors_before = false;
    bool fars_changed = false;
    size_t fars_snapshot = 0;
    bool late_far = false;          // a far factor added since the last solve whose older key had already left the window
    unsigned long long late_a = 0, late_b = 0;
    int late_n = 0;
    std::lock_guard<std::mutex> solve_lk(g->solve_mutex);
    {
        std::lock_guard<std::mutex> lk(g->graph_mutex);
        imus.swap(g->imu_queue);
        betweens.swap(g->staged_between);
        for (const auto& bt : betweens) g->set_band_end(bt.b, 1);   // these keys now carry their band factor (a second one goes to the far list)
        // (keys below the window can never be named again by a factor that is accepted: their marks are let go)
        while (!g->band_end.empty() && g->band_base < g->key_base + (uint64_t)g->lo) { g->band_end.pop_front(); g->band_base++; }
        // A far factor whose older key has left the window (as of the previous solve: lo / key_base change under solve_mutex,
        // which this thread holds) is no business of this list: one that was on the device then was marginalised with its key and
        // lives on in the engine (the list was replaced after that solve's marginalisations); one that was added SINCE that
        // solve never made it into the window -- late odometry, reported like a late band factor below
        {
            const uint64_t oldest = g->key_base + (uint64_t)g->lo;
            auto& fb = g->far_between;
            for (size_t i = fb.size() - (size_t)g->far_new; i < fb.size(); i++)
                if (fb[i].a < oldest) {
                    if (!late_far) { late_a = fb[i].a; late_b = fb[i].b; }
                    late_far = true;
                    late_n++;
                    for (size_t j = 0; j < g->staged_log.size(); j++)      // (dropped: it leaves graph()'s list as well)
                        if (g->staged_log[j].a == fb[i].a && g->staged_log[j].b == fb[i].b) { g->staged_log.erase(g->staged_log.begin() + (long)j); break; }
                }
            fb.erase(std::remove_if(fb.begin(), fb.end(), [&](const PendingBetween& f) { return f.a < oldest; }), fb.end());
        }
        fars = g->far_between;
        fars_snapshot = fars.size();
        far_new_before = g->far_new - late_n;      // (the late ones have just been erased and are reported below)
        g->far_new = 0;
        staged_before = g->staged_count - late_n;
        g->staged_count = 0;  // _graph->resize(0)
        log_before.swap(g->staged_log);
        priors_before = g->priors_staged;
        g->priors_staged = false;
        last_key = g->current_key;
        last_time = g->last_pose_time;
    }
    // Lock order is graph -> state everywhere (vf_reserve_node, vf_set_initial_state; GraphManager.cpp:54,60 -> 176), and the
    // reference's solve() has released _graphMutex before it takes _stateMutex (:104-117).  A solve that fails before the
    // optimisation has run must not lose what it took from the queues: every step up to vf_engine_iterate is idempotent
    // on the device (the same records, predictions and ranges are written again), so the snapshots go back to the FRONT
    // of the queues and the next vf_solve repeats them -- AFTER the state lock is released (requeue below), never
    // while it is held.  solve_mutex (taken first, by vf_solve only) keeps a second solver from snapshotting newer
    // entries between the failure and the give-back.
    bool requeue = false, late_band = false;
    uint64_t late_band_key = 0, late_band_a = 0;
    auto give_back = [&](int code) { requeue = true; return code; };
    auto locked = [&]() -> int {
    std::lock_guard<std::mutex> sl(g->state_mutex);  // :117
    int rc;
    // host-side validation first: a between factor whose source keyframe has already left the fixed-lag window (late
    // odometry) can never be added; it is dropped, the rest is given back, and the caller is told once
    {
        const uint64_t oldest = g->key_base + (uint64_t)g->lo;
        if (late_far)
            return give_back(gerr(VF_ERR_BAD_KEY, "between factor (%llu, %llu) dropped: key %llu left the fixed-lag window (oldest key %llu)",
                        late_a, late_b, late_a, (unsigned long long)oldest));
        for (size_t i = 0; i < betweens.size(); i++)
            if (betweens[i].a < oldest) {
                const unsigned long long a = betweens[i].a, b = betweens[i].b;
                late_band_key = betweens[i].b;          // (its end key carries no band factor after all: requeue clears the mark)
                late_band_a = betweens[i].a;
                late_band = true;
                betweens.erase(betweens.begin() + (long)i);
                staged_before--;
                return give_back(gerr(VF_ERR_BAD_KEY, "between factor (%llu, %llu) dropped: key %llu left the fixed-lag window (oldest key %llu)",
                            a, b, a, (unsigned long long)oldest));
            }
    }
    // whole-history mode (lag = 0, the reference's unbounded graph): the engine grows with the history
    if (g->opts.lag == 0 && (int)(last_key - g->key_base) + 1 > g->opts.capacity) {
        int want = g->opts.capacity;
        while ((int)(last_key - g->key_base) + 1 > want) want *= 2;
        if ((rc = vf_engine_grow(g->eng, want))) return give_back(rc);
        g->opts.capacity = want;
        lap("grow");
    }
    // fixed-lag mode: reclaim slots below the window when the new keyframes would not fit
    if (g->opts.lag > 0 && (int)(last_key - g->key_base) + 1 > g->opts.capacity) {
        const int shift = (g->lo / 64) * 64;
        if (shift <= 0) return give_back(gerr(VF_ERR_CAPACITY, "capacity %d too small for lag %d plus the keyframes added per solve", g->opts.capacity, g->opts.lag));
        if ((rc = vf_engine_compact(g->eng, shift))) return give_back(rc);
        g->lo -= shift;
        g->key_base += shift;
        if ((int)(last_key - g->key_base) + 1 > g->opts.capacity) return give_back(gerr(VF_ERR_CAPACITY, "capacity %d exhausted even after compaction", g->opts.capacity));
    }
    // fixed-lag window: marginalise the keyframes that fall out of the lag, one at a time, at the
    // linearisation of the previous solve (their factors have not changed since).  FIRST: the engine runs it on a second
    // stream, beside the preintegration / prediction / staging of the keyframes that arrive (vf_engine_set_async)
    const int last_slot = (int)(last_key - g->key_base);
    bool marginalised = false;
    while (g->opts.lag > 0 && last_slot + 1 - g->lo > g->opts.lag && (int)(g->solved_key - g->key_base) - g->lo >= 3) {
        if ((rc = vf_engine_marginalize(g->eng))) return give_back(rc);
        if ((rc = vf_engine_drop_oldest(g->eng))) return give_back(rc);
        g->lo++;      // (kept in step with the device: a keyframe that has been marginalised stays marginalised)
        marginalised = true;
    }
    const int lo = g->lo;
    lap("marginalize");
    // a far factor added since the last solve whose older key the marginalisations above have just moved out of the window never
    // reached the device: it is late odometry like any other (dropped, reported once, the rest given back)
    {
        const uint64_t oldest = g->key_base + (uint64_t)lo;
        bool dropped = false;
        unsigned long long da = 0, db = 0;
        for (size_t i = 0; i < fars.size();) {
            if (!fars[i].on_device && fars[i].a < oldest) {
                if (!dropped) { da = fars[i].a; db = fars[i].b; }
                dropped = true;
                for (size_t j = 0; j < log_before.size(); j++)
                    if (log_before[j].a == fars[i].a && log_before[j].b == fars[i].b) { log_before.erase(log_before.begin() + (long)j); break; }
                fars.erase(fars.begin() + (long)i);
                staged_before--;
                far_new_before--;
                fars_changed = true;
            } else i++;
        }
        if (dropped)
            return give_back(gerr(VF_ERR_BAD_KEY, "between factor (%llu, %llu) dropped: key %llu left the fixed-lag window in this solve (oldest key %llu)",
                                  da, db, da, (unsigned long long)oldest));
    }
    if (!imus.empty()) {
        // K0 on the device for all queued factors, then the initial values by IMU prediction
        // (GraphManager.cpp:150-160).  Keys are consecutive by construction.
        const int n = (int)imus.size();
        const uint64_t k0 = imus.front().key - g->key_base;
        // runs of factors cut from the IMU buffer (reserveNode) are preintegrated on the device in one K0 launch;
        // ready-made factors (addFactor) are staged as records
        for (int i = 0; i < n;) {
            int j = i;
            if (imus[i].staged) { i++; continue; }          // preintegrated at reserveNode time: the record is on the device
            const bool ready = !imus[i].record.empty();
            while (j < n && !imus[j].staged && imus[j].record.empty() == !ready) j++;
            const int cnt = j - i;
            if (ready) {
                std::vector<double> recs((size_t)cnt * VF_IMU_RECORD);
                for (int l = 0; l < cnt; l++) memcpy(&recs[(size_t)l * VF_IMU_RECORD], imus[i + l].record.data(), sizeof(double) * VF_IMU_RECORD);
                if ((rc = vf_engine_set_imu(g->eng, 0, (int)k0 + i, cnt, recs.data()))) return give_back(rc);
            } else {
                std::vector<int32_t> off(cnt + 1, 0);
                std::vector<double> steps, bias((size_t)cnt * 6);
                for (int l = 0; l < cnt; l++) {
                    off[l + 1] = off[l] + (int)(imus[i + l].steps.size() / 7);
                    steps.insert(steps.end(), imus[i + l].steps.begin(), imus[i + l].steps.end());
                    memcpy(&bias[(size_t)l * 6], imus[i + l].bias, sizeof(double) * 6);
                }
                if ((rc = vf_engine_preintegrate(g->eng, 0, (int)k0 + i, cnt, off.data(), steps.data(), bias.data(), &g->imu))) return give_back(rc);
            }
            i = j;
        }
        lap("preintegrate");
        // reference_compat: from the ESTIMATE of the last keyframe (_currentState, GraphManager.cpp:152), not its linearisation point
        if ((rc = (g->opts.reference_compat && g->solved_key > 0) ? vf_engine_predict_from_estimate(g->eng, 0, (int)k0, n)
                                                                    : vf_engine_predict(g->eng, 0, (int)k0, n))) return give_back(rc);
        lap("predict");
    }
    if (!betweens.empty()) {
        std::vector<int32_t> a(betweens.size()), b(betweens.size());
        std::vector<double> rec(betweens.size() * VF_BTW_RECORD);
        // the engine wants runs sorted by b
        std::vector<size_t> order(betweens.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        for (size_t i = 1; i < order.size(); i++)
            for (size_t j = i; j > 0 && betweens[order[j]].b < betweens[order[j - 1]].b; j--) std::swap(order[j], order[j - 1]);
        for (size_t i = 0; i < order.size(); i++) {
            a[i] = (int32_t)(betweens[order[i]].a - g->key_base);
            b[i] = (int32_t)(betweens[order[i]].b - g->key_base);
            memcpy(&rec[i * VF_BTW_RECORD], betweens[order[i]].rec, sizeof(double) * VF_BTW_RECORD);
        }
        if ((rc = vf_engine_set_between(g->eng, 0, (int)a.size(), a.data(), b.data(), rec.data()))) return give_back(rc);
        lap("set_between");
    }
    if (marginalised && (g->far_on_device || g->far_linear.load() > 0)) {
        // the engine has marginalised the far factors whose older key left together with that key (they are linear rows of its
        // own now): the entries that were on the device are replaced by what the engine's list holds now
        int cnt = 0;
        int32_t ea[VF_MAX_FAR_LIMIT], eb[VF_MAX_FAR_LIMIT];
        double erec[VF_MAX_FAR_LIMIT * VF_BTW_RECORD];
        if ((rc = vf_engine_get_extra_between(g->eng, 0, &cnt, ea, eb, erec, nullptr, nullptr, nullptr))) return give_back(rc);
        std::vector<PendingBetween> moved;
        for (int i = 0; i < cnt; i++) {
            PendingBetween f;
            f.a = (uint64_t)ea[i] + g->key_base;
            f.b = (uint64_t)eb[i] + g->key_base;
            memcpy(f.rec, erec + (size_t)i * VF_BTW_RECORD, sizeof(f.rec));
            f.on_device = true;
            moved.push_back(f);
        }
        for (const auto& f : fars)
            if (!f.on_device) moved.push_back(f);
        fars.swap(moved);
        fars_changed = true;
        int lin = 0;
        if ((rc = vf_engine_get_linear_far(g->eng, 0, &lin, nullptr))) return give_back(rc);
        g->far_linear.store(lin);
    }
    if (!fars.empty() || g->far_on_device) {
        // far between factors still inside the window, in window-local slots (the engine ignores one whose older keyframe
        // is below `lo`; the list is re-sent at every solve, so compaction and growth need no bookkeeping here)
        std::vector<int32_t> a, b;
        std::vector<double> rec;
        for (const auto& f : fars) {
            if (f.a < g->key_base + (uint64_t)lo || f.b > last_key) continue;
            a.push_back((int32_t)(f.a - g->key_base));
            b.push_back((int32_t)(f.b - g->key_base));
            rec.insert(rec.end(), f.rec, f.rec + VF_BTW_RECORD);
        }
        if ((rc = vf_engine_set_extra_between(g->eng, 0, (int)a.size(), a.data(), b.data(), rec.data()))) return give_back(rc);
        g->far_on_device = !a.empty();
        for (auto& f : fars)
            if (!(f.a < g->key_base + (uint64_t)lo || f.b > last_key)) { fars_changed = fars_changed || !f.on_device; f.on_device = true; }
        lap("set_extra");
    }
    if ((rc = vf_engine_set_range(g->eng, 0, lo, last_slot + 1))) return give_back(rc);
    lap("set_range");
    int fails = 0, flags = 0;
    if (g->opts.reference_compat) {
        if ((rc = vf_engine_isam_step(g->eng, g->opts.relin_threshold))) return rc;  // one ISAM2::update
        lap("isam_step");
        if ((rc = vf_engine_read_result(g->eng, 0, last_slot, 1, g->state, nullptr, nullptr, nullptr, nullptr, &flags))) return rc;   // calculateEstimate, :127,131-133
    } else {
        if ((rc = vf_engine_iterate(g->eng, g->opts.iterations))) return rc;   // (ISAM2::update + calculateEstimate: LM to convergence)
        lap("iterate(launch)");
        // the one synchronisation of the solve: the last state (:131-133), the failure count, what the staging kernels flagged
        if ((rc = vf_engine_read_result(g->eng, 0, last_slot, 0, g->state, nullptr, nullptr, nullptr, &fails, &flags))) return rc;
    }
    lap("read_result(sync)");
    // (asynchronous staging: a preintegrated covariance / a marginalisation pivot that was not positive definite shows here, after
    // the solve that used it -- with positive IMU covariances and a determined graph neither happens)
    if (flags & 1) return gerr(VF_ERR_NOT_SPD, "preintegrated covariance not positive definite");
    if (flags & 6) return gerr(VF_ERR_INDETERMINATE, "marginalisation: %s not positive definite", (flags & 4) ? "the far ends' block of the marginal" : "pivot block of the oldest keyframe");
    g->solved_key = last_key;
    for (auto& cb : g->callbacks)  // :135-138, on the solving thread, inside _stateMutex
        cb.first(cb.second, last_time, g->state, g->state + 4, g->state + 7, g->state + 10);
    // a full fixed-lag window marginalises its oldest keyframe at the next update, from the linearisation this solve leaves:
    // have the device compute that marginal prior now, behind the solve, instead of in front of the next one
    if (g->opts.lag > 0 && !g->opts.synchronous_staging && last_slot + 1 - g->lo >= g->opts.lag) (void)vf_engine_marginalize_ahead(g->eng);
    if (fails > 0 && fails >= g->opts.iterations && g->opts.iterations > 0)
        return gerr(VF_ERR_INDETERMINATE, "normal equations not positive definite in every LM trial (underdetermined graph?)");
    return VF_OK;
    };
    const int rc = locked();
    if (fars_changed) {
        // the entries snapshotted above are the front of the list (others only append, under graph_mutex): replace them by
        // their current form (the ones the engine has taken over are gone; on_device marks)
        std::lock_guard<std::mutex> lk(g->graph_mutex);
        if (g->far_between.size() >= fars_snapshot) {
            std::vector<PendingBetween> now(fars);
            now.insert(now.end(), g->far_between.begin() + (long)fars_snapshot, g->far_between.end());
            g->far_between.swap(now);
        }
    }
    if (requeue) {
        std::lock_guard<std::mutex> lk(g->graph_mutex);
        for (auto it = imus.rbegin(); it != imus.rend(); ++it) g->imu_queue.push_front(std::move(*it));
        g->staged_between.insert(g->staged_between.begin(), betweens.begin(), betweens.end());
        g->staged_count += staged_before;
        if (late_band)
            for (size_t j = 0; j < log_before.size(); j++)
                if (log_before[j].b == late_band_key && log_before[j].a == late_band_a) { log_before.erase(log_before.begin() + (long)j); break; }
        g->staged_log.insert(g->staged_log.begin(), log_before.begin(), log_before.end());
        g->priors_staged = g->priors_staged || priors_before;
        // the far factors added since the last successful solve are still "new" for the solve that repeats this one (its
        // late-odometry test looks at the new ones only), and a band factor dropped as late leaves its end key free again
        g->far_new += far_new_before;
        if (late_band) g->set_band_end(late_band_key, 0);
    }
    return rc;
}

int vf_get_state(vf_graph* g, double q[4], double t[3], double v[3], double bias[6]) {
    if (!g) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->state_mutex);
    if (q) memcpy(q, g->state, sizeof(double) * 4);
    if (t) memcpy(t, g->state + 4, sizeof(double) * 3);
    if (v) memcpy(v, g->state + 7, sizeof(double) * 3);
    if (bias) memcpy(bias, g->state + 10, sizeof(double) * 6);
    return VF_OK;
}

int vf_get_bias(vf_graph* g, double bias[6]) { return vf_get_state(g, nullptr, nullptr, nullptr, bias); }

int vf_graph_lm_stats(vf_graph* g, double* cost, int* accepted, int* rejected, int* solve_failures) {
    if (!g) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->state_mutex);
    return vf_engine_read_lm(g->eng, 0, cost, nullptr, accepted, rejected, solve_failures);
}

int vf_graph_solver_info(vf_graph* g, int* window_keyframes, int* refine_corrections, int* provisional_trials) {
    if (!g) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->state_mutex);
    if (window_keyframes) *window_keyframes = (int)(g->solved_key - g->key_base) + 1 - g->lo;
    int rc;
    if (refine_corrections && (rc = vf_engine_refine_count(g->eng, refine_corrections))) return rc;
    if (provisional_trials && (rc = vf_engine_read_excursions(g->eng, 0, provisional_trials, nullptr))) return rc;
    return VF_OK;
}

int vf_graph_incremental_info(vf_graph* g, long* updates, long* whole_window_updates, uint64_t* first_eliminated_key, uint64_t* last_substituted_key) {
    if (!g) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->state_mutex);
    int a = -1, b = -1;
    if (int rc = vf_engine_incremental_info(g->eng, 0, updates, whole_window_updates, &a, &b)) return rc;
    if (first_eliminated_key) *first_eliminated_key = a < 0 ? 0 : (uint64_t)a + g->key_base;
    if (last_substituted_key) *last_substituted_key = b < 0 ? 0 : (uint64_t)b + g->key_base;
    return VF_OK;
}

int vf_get_trajectory(vf_graph* g, uint64_t key0, int n, double* state16) {
    if (!g || !state16) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->state_mutex);
    if (n < 0 || key0 + (uint64_t)n > g->solved_key + 1) return gerr(VF_ERR_BAD_KEY, "keys [%llu, %llu) not solved yet", (unsigned long long)key0, (unsigned long long)(key0 + n));
    if (key0 < g->key_base) return gerr(VF_ERR_BAD_KEY, "key %llu has been compacted away (oldest retained key %llu)", (unsigned long long)key0, (unsigned long long)g->key_base);
    if (g->opts.reference_compat) return vf_engine_get_estimate(g->eng, 0, (int)(key0 - g->key_base), n, state16);
    return vf_engine_get_states(g->eng, 0, (int)(key0 - g->key_base), n, state16);
}

int vf_get_imu_factor(vf_graph* g, uint64_t key, double* rec190) {
    if (!g || !rec190) return gerr(VF_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g->state_mutex);
    if (key < 1 || key > g->solved_key || key < g->key_base + 1) return gerr(VF_ERR_BAD_KEY, "imu factor %llu not on the device", (unsigned long long)key);
    return vf_engine_get_imu(g->eng, 0, (int)(key - g->key_base), 1, rec190);
}

}  // extern "C"
