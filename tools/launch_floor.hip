// Back-to-back launch floor of this GPU (development tool): hipcc --offload-arch=gfx950 -O3 -o /tmp/floor tools/launch_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(float* p) {}
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * 64 + threadIdx.x; if (i < n) p[i] = p[i] + 1.0f; }
__global__ void k_rows(float* p, int B) { int e = blockIdx.x * 8 + (threadIdx.x & 7), b = threadIdx.x >> 3; float s = 0; for (int f = 0; f < 6; ++f) s += p[(size_t)(b * 6 + f) * B + e]; for (int f = 0; f < 6; ++f) p[(size_t)(b * 6 + f) * B + e] = s; }
int main() {
    float* p; hipMalloc(&p, 64 << 20); hipMemset(p, 0, 64 << 20);
    hipStream_t s; hipStreamCreate(&s);
    auto run = [&](const char* name, auto f) {
        for (int i = 0; i < 300; ++i) f();
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        const int K = 5000;
        for (int i = 0; i < K; ++i) f();
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
        printf("%-40s %6.2f us per launch\n", name, us);
    };
    run("empty, 1 block", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); });
    run("empty, 512 blocks", [&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(64), 0, s, p); });
    run("touch 32768 floats, 512 blocks", [&] { hipLaunchKernelGGL(k_touch, dim3(512), dim3(64), 0, s, p, 32768); });
    run("48 rows x 4096 envs read+write", [&] { hipLaunchKernelGGL(k_rows, dim3(512), dim3(64), 0, s, p, 4096); });
    return 0;
}
