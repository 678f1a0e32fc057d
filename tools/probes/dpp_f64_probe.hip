// Checks the semantics the generated pivot code relies on: v_fmac_f64_dpp with row_newbcast:n multiplies by the
// value lane n OF THE SAME 16-LANE ROW holds in src0 (gfx90a+; the only DPP control 64-bit ALU ops accept).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out) {
    const int lane = threadIdx.x;
    double acc = 1000.0 * lane, a = 1.0 + lane, b = 0.5;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4" : "+v"(acc) : "v"(a), "v"(b));
    out[lane] = acc;
}
int main() {
    double* d; hipMalloc(&d, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const double want = 1000.0 * l - (1.0 + (l / 16) * 16 + 3) * 0.5;
        if (h[l] != want) { bad++; printf("lane %d got %g want %g\n", l, h[l], want); }
    }
    printf("dpp probe: %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}
