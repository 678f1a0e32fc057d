/* Minimal C caller of libvilfusion.so through include/vilfusion.h: what a non-Python host (the reference's C++ node, a
 * cgo / JNI binding) does.  Feeds a vehicle at rest for one second -- IMU at 200 Hz, a keyframe every 50 ms with an
 * identity odometry factor between consecutive keyframes, plus one loop closure between keyframes 12 apart (a "far" factor:
 * GraphManager::addBetweenFactor takes any pair of keys, GraphManager.cpp:83-88) -- and prints the estimate after each solve.
 *
 *   gcc -I include examples/minimal.c -L vil_sensor_fusion_amd -lvilfusion -Wl,-rpath,$PWD/vil_sensor_fusion_amd -o /tmp/minimal
 *
 * Exit status: 0 = ran on a GPU and the estimate stayed at rest; 7 = no gfx950 device visible (the library has no CPU
 * path: VF_ERR_NO_DEVICE); anything else = failure. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "vilfusion.h"

static int n_callbacks = 0;
static void on_estimate(void* user, double time, const double q[4], const double t[3], const double v[3], const double bias[6]) {
    (void)user; (void)bias;
    n_callbacks++;
    printf("t = %.2f s  position (%+.2e %+.2e %+.2e) m  velocity (%+.2e %+.2e %+.2e) m/s  q_w %.12f\n", time, t[0], t[1], t[2], v[0], v[1],
           v[2], q[0]);
}

int main(void) {
    vf_imu_params imu = {1e-6, 1e-6, 1e-8, 1e-4, 1e-6, 1e-4}; /* config/carla/fusion_params.yaml:22-27 */
    vf_graph_opts opts;
    vf_graph_default_opts(&opts);
    opts.capacity = 256;
    opts.lag = 0;
    vf_graph* g = NULL;
    int rc = vf_create(&imu, &opts, &g);
    if (rc == VF_ERR_NO_DEVICE) {
        printf("no device: %s\n", vf_last_error());
        return 7;
    }
    if (rc) { printf("vf_create: %s\n", vf_last_error()); return 1; }
    vf_set_callback(g, on_estimate, NULL);
    const double acc[3] = {0.0, 0.0, 9.81}, gyro[3] = {0.0, 0.0, 0.0};     /* at rest, Z up (MakeSharedU) */
    const double q_id[4] = {1, 0, 0, 0}, t0[3] = {0, 0, 0};
    double cov[36];
    memset(cov, 0, sizeof(cov));
    for (int i = 0; i < 6; i++) cov[i * 6 + i] = 0.1;
    uint64_t prev = 0;
    double t = 0.0;
    for (int k = 1; k <= 20; k++) {
        for (int s = 0; s < 10; s++) { t += 0.005; if ((rc = vf_add_imu(g, t, acc, gyro))) goto fail; }
        uint64_t key = 0;
        if ((rc = vf_reserve_node(g, t, &key))) goto fail;
        if (prev && (rc = vf_add_between(g, prev, key, q_id, t0, cov))) goto fail;
        if (key == 15 && (rc = vf_add_between(g, 3, key, q_id, t0, cov))) goto fail;    /* a place revisited: keys 3 and 15 */
        prev = key;
        if ((rc = vf_solve(g))) goto fail;
    }
    {
        double q[4], p[3], v[3], b[6];
        vf_get_state(g, q, p, v, b);
        const double drift = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        printf("20 keyframes, %d callbacks, final drift %.3e m\n", n_callbacks, drift);
        vf_destroy(g);
        return (n_callbacks == 20 && drift < 1e-6) ? 0 : 2;
    }
fail:
    printf("error %d: %s\n", rc, vf_last_error());
    vf_destroy(g);
    return 1;
}
