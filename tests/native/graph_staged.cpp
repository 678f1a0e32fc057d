// Host logic of GraphManager::graph() behind the C ABI (vf_graph_get_staged) against the device-free engine double
// (tests/test_graph_threads.py builds and runs this): the staged NonlinearFactorGraph of the reference holds the three priors
// of GraphManager.cpp:27-35 until the first solve takes them, then every BetweenFactor<Pose3> in the order addBetweenFactor
// took it (GraphManager.cpp:83-88), and solve() empties it (:112-114).  test/UnitTests.cpp:200,222-233 reads its size, a
// factor's keys and measured().
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/vilfusion.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "line %d: %s failed (%s)\n", __LINE__, #cond, vf_last_error()); return 1; } } while (0)

int main() {
    vf_imu_params imu{1e-6, 1e-6, 1e-8, 1e-4, 1e-6, 1e-4};
    vf_graph_opts o;
    vf_graph_default_opts(&o);
    CHECK(o.struct_size == sizeof(o));
    o.capacity = 256;
    o.lag = 16;
    vf_graph* g = nullptr;
    CHECK(vf_create(&imu, &o, &g) == VF_OK);
    const double acc[3] = {0, 0, 9.81}, gyro[3] = {0, 0, 0};
    double cov[36] = {0};
    for (int i = 0; i < 6; i++) cov[i * 7] = i < 3 ? 0.01 : 0.1;
    double t = 0.0;
    auto node = [&]() { for (int s = 0; s < 3; s++) { t += 0.005; vf_add_imu(g, t, acc, gyro); } uint64_t k = 0; vf_reserve_node(g, t, &k); return k; };
    int kind = -1, staged = 0;
    uint64_t k1 = 9, k2 = 9;
    double q[4], tr[3], c36[36];
    // the three priors, in the order GraphManager.cpp:33-35 adds them
    vf_graph_staged(g, &staged, nullptr);
    CHECK(staged == 3);
    for (int i = 0; i < 3; i++) {
        CHECK(vf_graph_get_staged(g, i, &kind, &k1, &k2, q, tr, c36) == VF_OK);
        CHECK(kind == i && k1 == 0 && k2 == 0);
    }
    CHECK(vf_graph_get_staged(g, 0, &kind, nullptr, nullptr, q, tr, c36) == VF_OK);
    CHECK(q[0] == 1.0 && q[1] == 0.0 && tr[0] == 0.0 && std::fabs(std::sqrt(c36[0]) - 1e-6) < 1e-18 && std::fabs(std::sqrt(c36[35]) - 5e-5) < 1e-17);
    CHECK(vf_graph_get_staged(g, 3, &kind, &k1, &k2, q, tr, c36) == VF_ERR_BAD_KEY);
    // UnitTests.cpp:222-233: one between factor X(1) -> X(2), measured (1, 1, 1), rotation (0.5, 0.5, 0.5, 0.5)
    CHECK(node() == 1 && node() == 2);
    const double qm[4] = {1.0, 1.0, 1.0, 1.0}, tm[3] = {1.0, 1.0, 1.0};            // (handed in un-normalised: gtsam::Rot3(w, x, y, z) normalises)
    CHECK(vf_add_between(g, 1, 2, qm, tm, cov) == VF_OK);
    vf_graph_staged(g, &staged, nullptr);
    CHECK(staged == 4);
    CHECK(vf_graph_get_staged(g, 3, &kind, &k1, &k2, q, tr, c36) == VF_OK);
    CHECK(kind == 3 && k1 == 1 && k2 == 2 && tr[0] == 1.0 && tr[1] == 1.0 && tr[2] == 1.0);
    for (int i = 0; i < 4; i++) CHECK(std::fabs(q[i] - 0.5) < 1e-15);
    CHECK(memcmp(c36, cov, sizeof(cov)) == 0);
    // a far factor (wider than the band) is part of graph() like any other, in the order it was added
    for (int k = 3; k <= 9; k++) { const uint64_t key = node(); CHECK(vf_add_between(g, key - 1, key, qm, tm, cov) == VF_OK); }
    CHECK(vf_add_between(g, 2, 9, qm, tm, cov) == VF_OK);
    vf_graph_staged(g, &staged, nullptr);
    CHECK(staged == 3 + 1 + 7 + 1);
    CHECK(vf_graph_get_staged(g, staged - 1, &kind, &k1, &k2, nullptr, nullptr, nullptr) == VF_OK && kind == 3 && k1 == 2 && k2 == 9);
    CHECK(vf_graph_get_staged(g, 4, &kind, &k1, &k2, nullptr, nullptr, nullptr) == VF_OK && k1 == 2 && k2 == 3);
    // solve() takes everything: _graph->resize(0)
    CHECK(vf_solve(g) == VF_OK);
    vf_graph_staged(g, &staged, nullptr);
    CHECK(staged == 0 && vf_graph_get_staged(g, 0, &kind, &k1, &k2, q, tr, c36) == VF_ERR_BAD_KEY);
    const uint64_t key = node();
    CHECK(vf_add_between(g, key - 1, key, qm, tm, cov) == VF_OK);
    CHECK(vf_graph_get_staged(g, 0, &kind, &k1, &k2, q, tr, c36) == VF_OK && kind == 3 && k1 == key - 1 && k2 == key);   // no priors any more
    vf_destroy(g);
    printf("graph_staged ok\n");
    return 0;
}
