// How often can ONE wave issue v_mfma_f64_16x16x4f64, and does a second wave on the same SIMD fill the gaps?
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_f64_rate.hip -o /tmp/mfma_f64_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void __launch_bounds__(64) k(double* out, int n, unsigned long long* cyc) {
    d4_t acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = (d4_t){0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
// the same with a dependent chain of vector FMAs between the matrix instructions (does the vector unit run under them?)
__global__ void __launch_bounds__(64) k_mixed(double* out, int n, int with_mfma, unsigned long long* cyc) {
    d4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double x = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        if (with_mfma) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 16; j++) x = fma(x, b, a);     // 16 dependent v_fma_f64
        if (with_mfma) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 16; j++) x = fma(x, b, a);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = x + acc0[0] + acc1[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
// independent vector FMAs: what the vector unit gives ONE wave
template <int CHAINS>
__global__ void __launch_bounds__(64) k_valu(double* out, int n, unsigned long long* cyc) {
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = threadIdx.x * 1e-3 + c;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9 * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = fma(acc[c], a, b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAINS> void run_valu(int blocks, int n, double* out, unsigned long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_valu<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, n, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k_valu<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, n, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double f = (double)n * CHAINS;
    printf("%5d waves, %2d independent v_fma_f64 chains: %.3f ms, %.2f ns = %.1f ticks per FMA per wave, %.1f TFLOP/s\n", blocks, CHAINS, ms,
           ms * 1e6 / f, (double)c / f, blocks * f * 128 / (ms * 1e-3) * 1e-12);
}
template <int CHAINS> void run(int blocks, int n, double* out, unsigned long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, n, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(64), 0, 0, out, n, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = (double)n * CHAINS;
    printf("%5d waves, %d independent chains: %.3f ms, %.1f ns per MFMA per wave, %.1f s_memtime ticks per MFMA, %.1f TFLOP/s\n", blocks, CHAINS, ms,
           ms * 1e6 / mf, (double)c / mf, blocks * mf * 2048 / (ms * 1e-3) * 1e-12);
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8192 * 64 * 8); hipMalloc(&cyc, 8);
    const int n = 20000;
    for (int blocks : {1, 1024, 2048, 4096}) { run<1>(blocks, n, out, cyc); run<2>(blocks, n, out, cyc); run<4>(blocks, n, out, cyc); }
    for (int blocks : {1024, 2048, 4096}) { run_valu<1>(blocks, 200000, out, cyc); run_valu<4>(blocks, 100000, out, cyc); run_valu<16>(blocks, 50000, out, cyc); }
    for (int blocks : {1024, 2048}) for (int w = 0; w < 2; w++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(64), 0, 0, out, n, w, cyc); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(64), 0, 0, out, n, w, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%5d waves, 32 dependent v_fma_f64 per trip %s: %.3f ms = %.1f ns per trip\n", blocks, w ? "+ 2 MFMA" : "alone   ", ms, ms * 1e6 / n);
    }
    return 0;
}
