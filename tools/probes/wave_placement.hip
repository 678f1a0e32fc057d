// Where do the two waves of a 128-thread workgroup land (SIMD, slot), four 40 KB workgroups per CU?
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/wave_placement.hip -o tools/probes/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(128) k(unsigned* out, int spin) {
    __shared__ double S[5088];
    S[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_ID, all 32 bits
    double x = S[threadIdx.x];
    for (int i = 0; i < spin; i++) x = fma(x, 1.0000001, 1e-9);                  // stay resident while the others arrive
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 2 + (threadIdx.x >> 6)] = hw;
    if (x == 12345.678) out[0] = 0;
}
int main() {
    const int B = 1024;
    unsigned* d; hipMalloc(&d, B * 2 * 4);
    hipLaunchKernelGGL(k, dim3(B), dim3(128), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 2);
    hipMemcpy(h.data(), d, B * 2 * 4, hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (+ xcc in another register)
    std::map<unsigned, std::vector<int>> per_simd;   // key: (se, sh, cu, simd) -> roles (0 = wave 0, 1 = wave 1)
    int same_simd = 0, pair_hist[4][4] = {};
    for (int b = 0; b < B; b++) {
        const unsigned a = h[2 * b], c = h[2 * b + 1];
        const int s0 = (a >> 4) & 3, s1 = (c >> 4) & 3;
        pair_hist[s0][s1]++;
        if (s0 == s1) same_simd++;
        for (int wv = 0; wv < 2; wv++) { const unsigned x = h[2 * b + wv]; per_simd[(x >> 4) & 0xfff].push_back(wv); }
        if (b < 8) printf("wg %d: wave0 simd %d slot %u cu %u se %u | wave1 simd %d slot %u cu %u se %u\n", b, s0, a & 15, (a >> 8) & 15, (a >> 13) & 7, s1, c & 15, (c >> 8) & 15, (c >> 13) & 7);
    }
    printf("(simd of wave 0, simd of wave 1) histogram:\n");
    for (int i = 0; i < 4; i++) printf("  %4d %4d %4d %4d\n", pair_hist[i][0], pair_hist[i][1], pair_hist[i][2], pair_hist[i][3]);
    int hist[3][5] = {};
    for (auto& kv : per_simd) { int n0 = 0; for (int r : kv.second) n0 += r == 0; const int n = (int)kv.second.size(); if (n <= 4) hist[n0 > 2 ? 2 : n0][n]++; }
    printf("per (se, cu, simd) key: waves resident n, of which wave-0s n0 -> count\n");
    for (int n0 = 0; n0 < 3; n0++) for (int n = 0; n < 5; n++) if (hist[n0][n]) printf("  n = %d, wave-0s = %d: %d\n", n, n0, hist[n0][n]);
    return 0;
}
