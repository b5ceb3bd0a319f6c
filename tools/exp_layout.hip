// Development microbenchmark: what does the row-strided SoA layout [rows][B] cost against a tile-major layout
// [tile][rows][64] when every wave reads R rows of its 64 envs and writes them back?  (One wave per workgroup, like the
// one-lane-per-env kernels.)   hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_layout tools/exp_layout.hip && /tmp/exp_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R, bool TILED, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void touch(float* __restrict__ st, size_t B, int per_xcd) {
    const int b = blockIdx.x;
    const int tile = (b & 7) * per_xcd + (b >> 3);       // XCD-aware tile map, as in the step kernels
    const size_t e = (size_t)tile * 64 + threadIdx.x;
    if (e >= B) return;
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = TILED ? st[((size_t)tile * R + r) * 64 + threadIdx.x] : st[(size_t)r * B + e];
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) acc += v[r];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float o = v[r] + acc * 1e-9f;
        if (TILED) st[((size_t)tile * R + r) * 64 + threadIdx.x] = o; else st[(size_t)r * B + e] = o;
    }
}

template <int R, bool TILED, int WAVES>
void run(float* st, size_t B, const char* name) {
    const int tiles = (int)((B + 63) / 64);
    const int grid = ((tiles + 7) / 8) * 8;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((touch<R, TILED, WAVES>), dim3(grid), dim3(64), 0, 0, st, B, grid / 8);
    CHECK(hipEventRecord(a));
    const int n = 20;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((touch<R, TILED, WAVES>), dim3(grid), dim3(64), 0, 0, st, B, grid / 8);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / n, bytes = 2.0 * R * 4.0 * (double)B;
    printf("%-34s B %8zu R %3d waves/SIMD %d: %8.1f us  %6.2f TB/s\n", name, B, R, WAVES, us, bytes / us / 1e6);
}

int main() {
    const size_t BMAX = (size_t)1 << 22;
    float* st; CHECK(hipMalloc(&st, BMAX * 96 * sizeof(float))); CHECK(hipMemset(st, 0, BMAX * 96 * sizeof(float)));
    // is it the power-of-two row stride?  same kernels, row stride = B = 2^k + padding
    for (size_t B : {((size_t)1 << 20) + 64, ((size_t)1 << 20) + 64 * 33, ((size_t)1 << 20) + 64 * 1057, ((size_t)1 << 22) - 64 * 1057, (size_t)1000000 / 64 * 64, (size_t)4000000 / 64 * 64}) {
        run<84, false, 3>(st, B, "SoA rows [84][B], padded B");
        run<43, false, 8>(st, B, "SoA rows [43][B], padded B");
    }
    for (size_t B : {(size_t)1 << 20, (size_t)1 << 22}) {
        run<43, false, 3>(st, B, "SoA rows [43][B]");
        run<43, true, 3>(st, B, "tile-major [tile][43][64]");
        run<43, false, 4>(st, B, "SoA rows [43][B]");
        run<43, true, 4>(st, B, "tile-major [tile][43][64]");
        run<43, false, 8>(st, B, "SoA rows [43][B]");
        run<43, true, 8>(st, B, "tile-major [tile][43][64]");
        run<84, false, 3>(st, B, "SoA rows [84][B]");
        run<84, true, 3>(st, B, "tile-major [tile][84][64]");
    }
    return 0;
}
