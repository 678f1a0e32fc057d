/* A time-sharded window driven from C: the two collectives of every solve are issued by the library
 * itself through RCCL (vf_shard_iterate), the communicator is this program's.  One rank here -- the box has one GPU --
 * so the collectives are RCCL calls over a one-rank communicator; with N ranks every process does exactly this with its
 * own rank / device (ncclCommInitRank instead of ncclCommInitAll).
 *
 * The window: 160 keyframes of a vehicle at rest (IMU at 200 Hz: gravity only; identity odometry between consecutive
 * keyframes), initial values disturbed by a few centimetres so that LM has work to do; the same window in an unsharded
 * engine (vf_engine_iterate) is the check: the sharded path must end with the same bits.
 *
 *   gcc -std=c99 -I include examples/sharded.c -L vil_sensor_fusion_amd -lvilfusion -L /opt/rocm/lib -lrccl -lm \
 *       -Wl,-rpath,$PWD/vil_sensor_fusion_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/sharded
 *
 * Exit status: 0 = ran and the two engines agree bit for bit; 7 = no device; anything else = failure. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the three RCCL entry points this program needs, as <rccl/rccl.h> declares them (that header pulls in the HIP runtime's
 * C++ headers; a plain C host keeps to the C ABI: ncclComm_t is an opaque pointer, ncclResult_t an int, 0 = ncclSuccess) */
typedef void* ncclComm_t;
int ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
int ncclCommDestroy(ncclComm_t comm);
#define ncclSuccess 0

#include "vilfusion.h"

#define N 160
#define CHUNKS 4

static int load(vf_engine* e) {
    const vf_imu_params imu = {1e-6, 1e-6, 1e-8, 1e-4, 1e-6, 1e-4};
    int32_t off[N];
    static double steps[(N - 1) * 10 * 7], bias[(N - 1) * 6], btw[(N - 1) * VF_BTW_RECORD], states[N * VF_STATE_DIM];
    int32_t a[N - 1], b[N - 1];
    memset(bias, 0, sizeof(bias));
    memset(btw, 0, sizeof(btw));
    for (int k = 0; k < N - 1; k++) {
        off[k] = 10 * k;
        for (int s = 0; s < 10; s++) {
            double* st = steps + (size_t)(10 * k + s) * 7;
            st[0] = 0.005; st[1] = 0.0; st[2] = 0.0; st[3] = 9.81; st[4] = st[5] = st[6] = 0.0;
        }
        a[k] = k; b[k] = k + 1;
        double* r = btw + (size_t)k * VF_BTW_RECORD;
        r[0] = 1.0;                                                     /* identity rotation, zero translation */
        for (int i = 0, o = 7; i < 6; o += 6 - i, i++) r[o] = 1.0 / sqrt(0.1);   /* R = I / sqrt(cov), packed upper */
    }
    off[N - 1] = 10 * (N - 1);
    double prior[VF_PRIOR_RECORD] = {1, 0, 0, 0};
    const double sig[15] = {1e-6, 1e-6, 1e-6, 5e-5, 5e-5, 5e-5, 1e-5, 1e-5, 1e-5, 1e-7, 1e-7, 1e-7, 1e-7, 1e-7, 1e-7};   /* GraphManager.cpp:27-31 */
    memcpy(prior + 16, sig, sizeof(sig));
    memset(states, 0, sizeof(states));
    for (int k = 0; k < N; k++) {
        double* x = states + (size_t)k * VF_STATE_DIM;
        x[0] = 1.0;
        if (k) { x[4] = 0.03 * sin(0.11 * k); x[5] = 0.02 * cos(0.07 * k); x[6] = 0.01 * sin(0.05 * k); x[7] = 0.01 * sin(0.2 * k); }
    }
    int rc;
    if ((rc = vf_engine_preintegrate(e, 0, 1, N - 1, off, steps, bias, &imu))) return rc;
    if ((rc = vf_engine_set_between(e, 0, N - 1, a, b, btw))) return rc;
    if ((rc = vf_engine_set_prior(e, 0, 0, prior))) return rc;
    if ((rc = vf_engine_set_states(e, 0, 0, N, states))) return rc;
    return vf_engine_set_range(e, 0, 0, N);
}

int main(void) {
    int ndev = 0;
    if (vf_device_count(&ndev) != VF_OK || ndev < 1) { printf("no device: %s\n", vf_last_error()); return 7; }
    vf_engine_opts o;
    vf_engine_default_opts(&o);
    o.windows = 1; o.capacity = N; o.chunks = CHUNKS;
    vf_engine *whole = NULL, *shard = NULL;
    int rc;
    if ((rc = vf_engine_create(&o, &whole)) || (rc = vf_engine_create(&o, &shard))) { printf("create: %s\n", vf_last_error()); return rc == VF_ERR_NO_DEVICE ? 7 : 1; }
    if ((rc = load(whole)) || (rc = load(shard))) { printf("load: %s\n", vf_last_error()); return 1; }
    ncclComm_t comm;
    int dev = 0;
    if (ncclCommInitAll(&comm, 1, &dev) != ncclSuccess) { printf("ncclCommInitAll failed\n"); return 3; }
    long so, sc, stot, dc;
    vf_shard_exchange_plan(1, N, CHUNKS, 0, 1, &so, &sc, &stot, &dc);
    printf("one rank of one: all-gathers %ld of %ld separator doubles at offset %ld, all-reduces %ld increments\n", sc, stot, so, dc);
    if ((rc = vf_engine_iterate(whole, 4))) { printf("iterate: %s\n", vf_last_error()); return 1; }
    if ((rc = vf_engine_set_shard(shard, 0, 1)) || (rc = vf_shard_iterate(shard, (void*)comm, 4))) { printf("shard: %s\n", vf_last_error()); return 1; }
    static double xa[N * VF_STATE_DIM], xb[N * VF_STATE_DIM];
    double ca, cb;
    int acc_a, acc_b, fa, fb;
    if ((rc = vf_engine_get_states(whole, 0, 0, N, xa)) || (rc = vf_engine_get_states(shard, 0, 0, N, xb)) ||
        (rc = vf_engine_read_lm(whole, 0, &ca, NULL, &acc_a, NULL, &fa)) || (rc = vf_engine_read_lm(shard, 0, &cb, NULL, &acc_b, NULL, &fb))) {
        printf("read: %s\n", vf_last_error());
        return 1;
    }
    double worst = 0.0, moved = 0.0;
    for (int i = 0; i < N * VF_STATE_DIM; i++) { const double d = fabs(xa[i] - xb[i]); if (d > worst) worst = d; }
    for (int k = 1; k < N; k++) { const double d = fabs(xa[k * VF_STATE_DIM + 4]); if (d > moved) moved = d; }
    printf("unsharded: cost %.12e, %d accepted; through RCCL: cost %.12e, %d accepted; largest state difference %.3e; largest |x| left %.3e m\n",
           ca, acc_a, cb, acc_b, worst, moved);
    ncclCommDestroy(comm);
    vf_engine_destroy(whole);
    vf_engine_destroy(shard);
    return (worst == 0.0 && ca == cb && acc_a == acc_b && acc_a >= 1 && fa == 0 && fb == 0 && moved < 1e-3) ? 0 : 2;
}
