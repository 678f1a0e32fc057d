// Issue cost and dependent latency of the instructions K4's chain is made of, one wave on one SIMD (gfx950), in shader
// cycles (s_memtime).  Build + run: hipcc --offload-arch=gfx950 -O2 tools/probes/inst_timing.hip -o /tmp/inst_timing && /tmp/inst_timing
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory")
#define T1(i) do { asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); if (threadIdx.x == 0) out[i] = (double)(t1 - t0); } while (0)
__global__ void __launch_bounds__(64) k(double* out, double seed) {
    unsigned long long t0, t1;
    __shared__ double S[1024];
    const int lane = threadIdx.x;
    int scratch = lane;
    double a = seed + lane, b = 1.0000001, c = 0.5, d0 = 1, d1 = 2, d2 = 3, d3 = 4, d4 = 5, d5 = 6, d6 = 7, d7 = 8;
    S[lane] = a; S[lane + 64] = b;
    // 0: empty
    T0(); T1(0);
    // 1: 64 independent v_fma_f64 (8 accumulators round robin)
    T0();
    REP4(REP4(asm volatile("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b), "v"(c));))
    T1(1);
    // 2: 64 dependent v_fma_f64
    T0();
    REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(b), "v"(c));)
    T1(2);
    // 3: 64 independent v_fmac_f64_dpp (4 accumulators)
    T0();
    REP16(asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(c));)
    T1(3);
    // 4: 64 dependent v_fmac_f64_dpp (accumulator chain)
    T0();
    REP64(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d4) : "v"(a), "v"(c));)
    T1(4);
    // 5: 64 x (2 readlane + fma with SGPR), independent accumulators
    T0();
    REP16(asm volatile("v_readlane_b32 s20, %4, 3\n\tv_readlane_b32 s21, %5, 3\n\ts_nop 0\n\tv_fma_f64 %0, s[20:21], %6, %0\n\t"
                       "v_readlane_b32 s22, %4, 4\n\tv_readlane_b32 s23, %5, 4\n\ts_nop 0\n\tv_fma_f64 %1, s[22:23], %6, %1\n\t"
                       "v_readlane_b32 s24, %4, 5\n\tv_readlane_b32 s25, %5, 5\n\ts_nop 0\n\tv_fma_f64 %2, s[24:25], %6, %2\n\t"
                       "v_readlane_b32 s26, %4, 6\n\tv_readlane_b32 s27, %5, 6\n\ts_nop 0\n\tv_fma_f64 %3, s[26:27], %6, %3"
                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    T1(5);
    // 6: 16 dependent v_rsq_f64
    T0();
    REP16(asm volatile("v_rsq_f64 %0, %0" : "+v"(d5));)
    T1(6);
    // 7: 16 independent v_rsq_f64
    T0();
    REP4(asm volatile("v_rsq_f64 %0, %4\n\tv_rsq_f64 %1, %4\n\tv_rsq_f64 %2, %4\n\tv_rsq_f64 %3, %4" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a));)
    T1(7);
    // 8: 16 x LDS write -> read round trip (dependent through memory), same lane
    T0();
    REP16(asm volatile("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(d6) : "v"(lane * 8));)
    T1(8);
    // 9: 16 dependent mfma f64 16x16x4 (same accumulator)
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
        T0();
        REP16(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);)
        asm volatile("s_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, %1" : "=v"(scratch) : "v"(__double2loint(acc[0])));
        T1(9);
        // 10: 48 mfma on 3 accumulators interleaved
        T0();
        REP16(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);)
        asm volatile("s_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, %1" : "=v"(scratch) : "v"(__double2loint(acc[0] + acc2[0] + acc3[0])));
        T1(10);
        d7 += acc[1] + acc2[2] + acc3[3];
    }
    // 11: 64 v_mul_f64 independent
    T0();
    REP16(asm volatile("v_mul_f64 %0, %4, %5\n\tv_mul_f64 %1, %4, %5\n\tv_mul_f64 %2, %4, %5\n\tv_mul_f64 %3, %4, %5" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(c));)
    T1(11);
    // 12: 64 accvgpr write+read pairs
    T0();
    REP64(asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(scratch) :: "a0");)
    T1(12);
    // 13: 64 s_nop 0
    T0();
    REP64(asm volatile("s_nop 0");)
    T1(13);
    // 14: 64 dependent v_readlane -> v_fma chain (the pivot chain's shape: scalar from a VALU result, used by the next VALU)
    T0();
    REP16(asm volatile("v_readlane_b32 s20, %1, 3\n\tv_readlane_b32 s21, %2, 3\n\ts_nop 0\n\tv_fma_f64 %0, s[20:21], %3, %0\n\t" : "+v"(d1) : "v"(__double2loint(d1)), "v"(__double2hiint(d1)), "v"(c) : "s20", "s21");)
    T1(14);
    // 15: 16 x ds_read_b64 x 15 (one panel row) then wait
    T0();
    REP16(asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(lane * 8 * 15 % 4096));)
    T1(15);
    if (lane == 63) out[31] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + scratch;
}
int main() {
    double* d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.5);
    double h[32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"empty", "64 indep v_fma_f64", "64 dep v_fma_f64", "64 indep v_fmac_f64_dpp", "64 dep v_fmac_f64_dpp", "64 x (2 readlane + nop + fma sgpr) indep",
                           "16 dep v_rsq_f64", "16 indep v_rsq_f64", "16 x lds write->read", "16 dep mfma_f64_16x16x4", "48 mfma 3 accumulators", "64 indep v_mul_f64",
                           "64 x accvgpr write+read", "64 s_nop 0", "16 x dep (2 readlane+nop+fma)", "16 x (4 ds_read_b64 + wait)"};
    const int counts[] = {1, 64, 64, 64, 64, 64, 16, 16, 16, 16, 48, 64, 64, 64, 16, 16};
    for (int i = 0; i < 16; i++) printf("%-44s %8.0f cycles  -> %6.1f per item (net of empty)\n", names[i], h[i], (h[i] - h[0]) / counts[i]);
    return 0;
}
