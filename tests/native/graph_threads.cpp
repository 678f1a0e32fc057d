// Two-lock discipline of the GraphManager surface under ThreadSanitizer (tests/test_graph_threads.py builds and runs this
// with vf_graph.cpp + fake_engine.cpp, -fsanitize=thread).  The reference's contract: every public method thread-safe
// (GraphManager.h:4-6), lock order _graphMutex -> _stateMutex (GraphManager.cpp:54,60 -> 176), solve() releases the graph
// lock before it takes the state lock (:104-117).  A solve that FAILS puts its snapshot back into the queues: that path
// must not take the graph lock while it still holds the state lock.
#include <atomic>
#include <cstdio>
#include <thread>

#include "../../include/vilfusion.h"

extern std::atomic<int> fake_fail_preintegrate;
extern std::atomic<long> fake_iterates;

int main() {
    vf_imu_params imu{1e-6, 1e-6, 1e-8, 1e-4, 1e-6, 1e-4};
    vf_graph_opts o;
    vf_graph_default_opts(&o);
    o.capacity = 1 << 20;
    vf_graph* g = nullptr;
    if (vf_create(&imu, &o, &g)) { fprintf(stderr, "create failed: %s\n", vf_last_error()); return 2; }
    const double acc[3] = {0, 0, 9.81}, gyro[3] = {0, 0, 0};
    std::atomic<bool> stop{false};
    std::atomic<long> solves{0}, failed{0}, nodes{0};
    const double eye[36] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1};
    const double q[4] = {1, 0, 0, 0}, t3[3] = {0.1, 0, 0};
    // ingestion thread: IMU samples, nodes, between factors (what the sensor callbacks do)
    std::thread ingest([&] {
        double t = 0.0;
        uint64_t prev = 0;
        for (int k = 0; k < 4000 && !stop.load(); k++) {
            for (int s = 0; s < 3; s++) { t += 0.005; vf_add_imu(g, t, acc, gyro); }
            uint64_t key = 0;
            if (vf_reserve_node(g, t, &key) == VF_OK) {
                nodes++;
                if (prev) vf_add_between(g, prev, key, q, t3, eye);
                prev = key;
            }
            double b[6];
            vf_get_bias(g, b);
            int st, qu;
            vf_graph_staged(g, &st, &qu);
        }
        stop = true;
    });
    // solving thread: every other solve fails inside the state lock and gives its queues back
    std::thread solver([&] {
        long i = 0;
        while (!stop.load()) {
            fake_fail_preintegrate = (int)(i++ & 1);
            if (vf_solve(g) == VF_OK) solves++; else failed++;
        }
        fake_fail_preintegrate = 0;
        vf_solve(g);
    });
    // a second solver (the reference allows solve() from any thread)
    std::thread solver2([&] { while (!stop.load()) { vf_solve(g); std::this_thread::yield(); } });
    ingest.join();
    solver.join();
    solver2.join();
    int staged = -1, queued = -1;
    vf_graph_staged(g, &staged, &queued);
    uint64_t key = 0;
    double tt = 0;
    vf_most_recent_pose_time(g, &tt, &key);
    printf("nodes %ld solves %ld failed %ld queued-after %d staged-after %d last key %llu\n", nodes.load(), solves.load(), failed.load(), queued, staged, (unsigned long long)key);
    vf_destroy(g);
    // nothing may be lost: after the final successful solve the queues are empty, and every node got a key
    return (queued == 0 && staged == 0 && (long)key == nodes.load() && failed.load() > 0) ? 0 : 1;
}
