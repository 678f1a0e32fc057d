// Host logic of the far-factor routing in vf_graph.cpp against the device-free engine double (tests/test_graph_threads.py
// builds and runs this): GraphManager::addBetweenFactor takes any pair of keys (GraphManager.cpp:83-88); a factor the band
// cannot hold -- wider than VF_MAX_BANDWIDTH, or a second one ending at a key -- goes to the far list, counts as a staged
// factor, reaches the engine through vf_engine_set_extra_between at every solve while both keys are in the window, and the
// list is bounded by vf_graph_opts.max_far_factors (VF_MAX_FAR_LIMIT by default).
#include <atomic>
#include <cstdio>

#include "../../include/vilfusion.h"

extern std::atomic<int> fake_band_n, fake_extra_n, fake_extra_calls, fake_extra_a0, fake_extra_b0;

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "line %d: %s failed (%s)\n", __LINE__, #cond, vf_last_error()); return 1; } } while (0)

int main() {
    vf_imu_params imu{1e-6, 1e-6, 1e-8, 1e-4, 1e-6, 1e-4};
    vf_graph_opts o;
    vf_graph_default_opts(&o);
    o.capacity = 256;
    o.max_far_factors = VF_MAX_EXTRA;      // (a handle's default is VF_MAX_FAR_LIMIT)
    o.lag = 16;
    vf_graph* g = nullptr;
    CHECK(vf_create(&imu, &o, &g) == VF_OK);
    const double acc[3] = {0, 0, 9.81}, gyro[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}, t3[3] = {0.1, 0, 0};
    double eye[36] = {0};
    for (int i = 0; i < 6; i++) eye[i * 7] = 1.0;
    double t = 0.0;
    auto node = [&]() { for (int s = 0; s < 3; s++) { t += 0.005; vf_add_imu(g, t, acc, gyro); } uint64_t k = 0; vf_reserve_node(g, t, &k); return k; };
    for (int k = 1; k <= 12; k++) { const uint64_t key = node(); if (key > 1) CHECK(vf_add_between(g, key - 1, key, q, t3, eye) == VF_OK); }
    int staged = 0, queued = 0;
    vf_graph_staged(g, &staged, &queued);
    CHECK(staged == 3 + 11);
    CHECK(vf_add_between(g, 2, 9, q, t3, eye) == VF_OK);         // span 7: far
    CHECK(vf_add_between(g, 5, 6, q, t3, eye) == VF_OK);         // key 6 already ends a staged band factor: far
    vf_graph_staged(g, &staged, &queued);
    CHECK(staged == 3 + 11 + 2);
    CHECK(vf_solve(g) == VF_OK);
    CHECK(fake_band_n.load() == 11 && fake_extra_n.load() == 2 && fake_extra_a0.load() == 2 && fake_extra_b0.load() == 9);
    vf_graph_staged(g, &staged, &queued);
    CHECK(staged == 0 && queued == 0);
    CHECK(vf_add_between(g, 10, 11, q, t3, eye) == VF_OK);       // key 11 got its band factor in the last solve: far
    CHECK(vf_solve(g) == VF_OK);
    CHECK(fake_extra_n.load() == 3);                              // the list persists: re-sent at every solve
    for (int i = 0; i < 5; i++) CHECK(vf_add_between(g, 1 + i, 8 + (uint64_t)i % 4, q, t3, eye) == VF_OK);
    CHECK(vf_add_between(g, 3, 12, q, t3, eye) == VF_ERR_CAPACITY);   // the ninth
    CHECK(vf_add_between(g, 12, 3, q, t3, eye) == VF_ERR_BAD_KEY);
    // keys leave the fixed-lag window (lag 16): a far factor whose older key is marginalised is TRANSPORTED to the next keyframe by
    // the engine (vf_engine_drop_oldest) and the GraphManager's entry follows -- the factor (2, 9) is (3, 9) after key 2 has
    // left, (4, 9) after key 3 ... -- until it reaches its own end key and is dropped
    int seen_moved = 0;
    for (int k = 13; k <= 60; k++) {
        const uint64_t key = node();
        CHECK(vf_add_between(g, key - 1, key, q, t3, eye) == VF_OK);
        CHECK(vf_solve(g) == VF_OK);
        if (key == 20) {            // window = keys [5, 20]: every older key so far has been transported up to 5 or ended before
            CHECK(fake_extra_n.load() >= 1 && fake_extra_a0.load() >= 0);
            seen_moved = 1;
        }
    }
    CHECK(seen_moved == 1 && fake_extra_n.load() == 0);
    const int calls = fake_extra_calls.load();
    node();
    CHECK(vf_solve(g) == VF_OK);
    CHECK(fake_extra_calls.load() == calls);                      // nothing far alive and nothing on the device: no call
    CHECK(vf_add_between(g, 50, 58, q, t3, eye) == VF_OK);        // room again
    CHECK(vf_solve(g) == VF_OK);
    CHECK(fake_extra_n.load() == 1);
    // late odometry on a far pair: the older key left the window before the factor was added -> the next solve drops it and
    // says so once (as for a late band factor), keeps everything else, and the solve after that is clean
    CHECK(vf_add_between(g, 20, 60, q, t3, eye) == VF_OK);
    const uint64_t k62 = node();
    CHECK(vf_add_between(g, k62 - 1, k62, q, t3, eye) == VF_OK);
    CHECK(vf_solve(g) == VF_ERR_BAD_KEY);
    vf_graph_staged(g, &staged, &queued);
    CHECK(staged == 1 && queued == 1);                             // given back: the band factor and the queued IMU factor
    CHECK(vf_solve(g) == VF_OK);
    CHECK(fake_extra_n.load() == 1);
    vf_destroy(g);
    // a handle made with the defaults holds VF_MAX_FAR_LIMIT of them (whole history: nothing ever leaves), then refuses; opts out of
    // range are refused at creation
    {
        vf_graph_opts d;
        vf_graph_default_opts(&d);
        CHECK(d.max_far_factors == 0);
        d.capacity = 256;
        vf_graph* h = nullptr;
        CHECK(vf_create(&imu, &d, &h) == VF_OK);
        t = 0.0;
        auto node2 = [&]() { for (int s = 0; s < 3; s++) { t += 0.005; vf_add_imu(h, t, acc, gyro); } uint64_t k = 0; vf_reserve_node(h, t, &k); return k; };
        for (int k = 1; k <= 80; k++) { const uint64_t key = node2(); if (key > 1) CHECK(vf_add_between(h, key - 1, key, q, t3, eye) == VF_OK); }
        for (int i = 0; i < VF_MAX_FAR_LIMIT; i++) CHECK(vf_add_between(h, 1 + (uint64_t)i, 40 + (uint64_t)i, q, t3, eye) == VF_OK);
        CHECK(vf_add_between(h, 2, 79, q, t3, eye) == VF_ERR_CAPACITY);
        CHECK(vf_solve(h) == VF_OK);
        CHECK(fake_extra_n.load() == VF_MAX_FAR_LIMIT);
        vf_destroy(h);
        d.max_far_factors = VF_MAX_FAR_LIMIT + 1;
        CHECK(vf_create(&imu, &d, &h) == VF_ERR_INVALID);
        d.max_far_factors = -1;
        CHECK(vf_create(&imu, &d, &h) == VF_ERR_INVALID);
    }
    printf("far-factor routing ok\n");
    return 0;
}
