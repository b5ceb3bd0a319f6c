// exec_skip.hip — does a wave64 VALU instruction get cheaper when only part of the wave is active (EXEC covering 32 / 16 / 8
// lanes)?  One wave per SIMD, a dependent v_fma_f32 chain and an independent one, cycles per instruction by active lanes.
// build: hipcc --offload-arch=gfx950 -O3 -o exec_skip exec_skip.hip ; run: ./exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int ITERS = 4000;
template <bool DEP>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, int active, int lo, unsigned long long mask = 0) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float m = 0.9999999f, c = 1e-7f;
    unsigned long long t0 = 0, t1 = 0;
    if (mask ? ((mask >> threadIdx.x) & 1ull) != 0 : ((int)threadIdx.x >= lo && (int)threadIdx.x < lo + active)) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < ITERS; ++i) {
            if (DEP) asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                                  "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(m), "v"(c));
            else asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                              "v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));
        }
        t1 = __builtin_readcyclecounter();
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (mask ? threadIdx.x == (unsigned)__builtin_ctzll(mask) : (int)threadIdx.x == lo) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const int blocks = 1024;   // one wave per SIMD
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 64 * 4 * 4); hipMalloc(&cyc, blocks * 8 * 4);
    std::vector<unsigned long long> h(blocks * 4);
    for (int dep = 0; dep < 2; ++dep)
        for (int lo : {0, 32, 48})
            for (int active : {64, 32, 16, 8, 1}) {
                if (lo + active > 64) continue;
                for (int rep = 0; rep < 2; ++rep) {
                    if (dep) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(64), 0, 0, out, cyc, active, lo);
                    else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(64), 0, 0, out, cyc, active, lo);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.end());
                printf("%s lanes [%2d, %2d): %.2f cycles per instruction (median wave)\n", dep ? "dependent  " : "independent", lo, lo + active, (double)h[blocks / 2] / (ITERS * 8.0));
            }
    struct { const char* name; unsigned long long m; } pats[] = {
        {"every 8th lane (8 lanes)", 0x0101010101010101ull}, {"lanes 6 mod 8 and 7 mod 8 (16)", 0xC0C0C0C0C0C0C0C0ull}, {"one lane per quarter (4)", 0x0001000100010001ull},
        {"one lane in quarters 0,1 (2)", 0x0000000000010001ull}, {"one lane in quarters 0,2 (2)", 0x0000000100000001ull}, {"lanes 0-16 (17)", 0x1FFFFull}, {"lanes 0-15 + 32 (17)", 0x10000FFFFull},
        {"lanes 0-23 (24)", 0xFFFFFFull}, {"lanes 0-15 and 32-47 (32)", 0x0000FFFF0000FFFFull}, {"lanes 0-7 of each quarter (32)", 0x00FF00FF00FF00FFull}, {"quarters 0,1,2 one lane each (3)", 0x0000000100010001ull},
        {"lane 0 and lane 63 (2)", 0x8000000000000001ull}, {"lanes 0-15 + lane 16 + lane 32 (18)", 0x10001FFFFull}};
    // wall clock and waves per SIMD: is the 14 cycles an issue cost (other waves cannot use the slots) or a latency?
    for (int W : {1, 2, 4}) for (unsigned long long m : {~0ull, 0x0101010101010101ull}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<true>, dim3(blocks * W), dim3(64), 0, 0, out, cyc, 0, 0, m); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k<true>, dim3(blocks * W), dim3(64), 0, 0, out, cyc, 0, 0, m); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("dependent   W=%d waves/SIMD mask %016llx: %.2f cycles per instruction per wave, kernel %.1f us = %.2f ns per instruction per wave\n", W, m, (double)h[blocks / 2] / (ITERS * 8.0), ms * 1e3, ms * 1e6 / (ITERS * 8.0));
    }
    for (auto& p : pats) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(64), 0, 0, out, cyc, 0, 0, p.m); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("dependent   %-40s: %.2f cycles per instruction\n", p.name, (double)h[blocks / 2] / (ITERS * 8.0));
    }
    return 0;
}
