// Development experiment: do back-to-back launches in ONE stream overlap when the barrier bit of the dispatch packet is
// cleared (hipExtAnyOrderLaunch)?  512 single-wave blocks, each a dependent FMA chain of ~7 us; every 37th block runs
// 40 % longer (the straggler that decides a launch).  No inter-launch dependency in this test (pure timing).
// Result on MI355X / ROCm 7.2: 73.0 vs 73.5 us per launch — no overlap: hip_ext.h says the flag is not supported on GFX9xx.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/anyorder tools/ubench/anyorder.hip && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ __launch_bounds__(64) void chain(float* p, int n_base, int shift) {
    float x = p[blockIdx.x * 64 + threadIdx.x];
    const int n = ((blockIdx.x + shift) % 37 == 0) ? n_base + n_base * 2 / 5 : n_base;
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
    p[blockIdx.x * 64 + threadIdx.x] = x;
}
int main() {
    float* p; hipMalloc(&p, 512 * 64 * 4); hipMemset(p, 0, 512 * 64 * 4);
    hipStream_t s; hipStreamCreate(&s);
    const int n_base = 3400, K = 4000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < K; ++k) {
                const int shift = mode == 2 ? k : 0;   // mode 2: the straggler moves from launch to launch
                if (mode == 0) hipLaunchKernelGGL(chain, dim3(512), dim3(64), 0, s, p, n_base, shift);
                else hipExtLaunchKernelGGL(chain, dim3(512), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, p, n_base, shift);
            }
            hipStreamSynchronize(s);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
            printf("%s: %.2f us per launch\n", mode == 0 ? "in-order launches          " : mode == 1 ? "any-order, fixed straggler  " : "any-order, moving straggler ", us);
        }
    }
    return 0;
}
