// exec_skip3.hip — the same question on COMPILED code (no inline asm): a loop of contact-response-like arithmetic (IEEE sqrt, divide,
// fused multiply-adds, clamps, selects) under a divergent branch taken by 64 / 32 / 17 / 16 / 8 / 2 lanes.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, unsigned long long mask, int iters) {
    float x = 0.01f * threadIdx.x, y = 0.5f - x, vx = 0.3f, vy = -0.2f, ax = 0.0f, ay = 0.0f, px = 0.0f, py = 0.0f;
    unsigned long long t0 = 0, t1 = 0;
    if ((mask >> threadIdx.x) & 1ull) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            const float dx = 0.07f - x, dy = 0.03f - y;
            const float d2 = __builtin_fmaf(dx, dx, dy * dy);
            const float d = sqrtf(d2), inv = 1.0f / d;
            const float nx = dx * inv, ny = dy * inv;
            const float vn = __builtin_fmaf(vx, nx, vy * ny);
            if (vn < 0.0f) {
                const float q = 1.5f * vn * 0.5f;
                ax = __builtin_fmaf(q, nx, ax); ay = __builtin_fmaf(q, ny, ay);
                const float vt = __builtin_fmaf(vy, nx, -(vx * ny)) - 0.1f;
                const float lim = q * 0.4f;
                const float ft = __builtin_amdgcn_fmed3f(vt * 0.3f, lim, -lim);
                ax = __builtin_fmaf(-ft, ny, ax); ay = __builtin_fmaf(ft, nx, ay);
            }
            const float pc = 0.8f * (0.08f - d) * 0.5f;
            px = __builtin_fmaf(-pc, nx, px); py = __builtin_fmaf(-pc, ny, py);
            x = x + 1e-4f * ax; y = y + 1e-4f * ay; vx = vx + 1e-3f * px; vy = vy - 1e-3f * py;
        }
        t1 = __builtin_readcyclecounter();
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + y + vx + vy + ax + ay + px + py;
    if (threadIdx.x == (unsigned)__builtin_ctzll(mask)) cyc[blockIdx.x] = t1 - t0;
}
__device__ __forceinline__ void body(float& x, float& y, float& vx, float& vy, float& ax, float& ay, float& px, float& py) {
    const float dx = 0.07f - x, dy = 0.03f - y;
    const float d2 = __builtin_fmaf(dx, dx, dy * dy);
    const float d = sqrtf(d2), inv = 1.0f / d;
    const float nx = dx * inv, ny = dy * inv;
    const float vn = __builtin_fmaf(vx, nx, vy * ny);
    if (vn < 0.0f) {
        const float q = 1.5f * vn * 0.5f;
        ax = __builtin_fmaf(q, nx, ax); ay = __builtin_fmaf(q, ny, ay);
        const float vt = __builtin_fmaf(vy, nx, -(vx * ny)) - 0.1f;
        const float lim = q * 0.4f;
        const float ft = __builtin_amdgcn_fmed3f(vt * 0.3f, lim, -lim);
        ax = __builtin_fmaf(-ft, ny, ax); ay = __builtin_fmaf(ft, nx, ay);
    }
    const float pc = 0.8f * (0.08f - d) * 0.5f;
    px = __builtin_fmaf(-pc, nx, px); py = __builtin_fmaf(-pc, ny, py);
    x = x + 1e-4f * ax; y = y + 1e-4f * ay; vx = vx + 1e-3f * px; vy = vy - 1e-3f * py;
}
// K iterations under `mask`, then K iterations on all lanes, repeated: total iters each
__global__ __launch_bounds__(64) void kalt(float* out, unsigned long long* cyc, unsigned long long mask, int iters, int K) {
    float x = 0.01f * threadIdx.x, y = 0.5f - x, vx = 0.3f, vy = -0.2f, ax = 0.0f, ay = 0.0f, px = 0.0f, py = 0.0f;
    const bool in = (mask >> threadIdx.x) & 1ull;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters / K; ++i) {
        if (in) for (int q = 0; q < K; ++q) body(x, y, vx, vy, ax, ay, px, py);
        for (int q = 0; q < K; ++q) body(x, y, vx, vy, ax, ay, px, py);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x + y + vx + vy + ax + ay + px + py;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const int blocks = 1024, iters = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    std::vector<unsigned long long> h(blocks);
    struct { const char* n; unsigned long long m; } pats[] = {{"64 lanes", ~0ull}, {"32 lanes", 0xFFFFFFFFull}, {"17 lanes", 0x1FFFFull}, {"16 lanes", 0xFFFFull},
        {"16 lanes (2 per 8)", 0xC0C0C0C0C0C0C0C0ull}, {"8 lanes (1 per 8)", 0x0101010101010101ull}, {"2 lanes", 0x3ull}};
    for (auto& p : pats) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, cyc, p.m, iters); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%-22s %.1f cycles per iteration\n", p.n, (double)h[blocks / 2] / iters);
    }
    for (int K : {1, 2, 8, 64}) {
        double r[2];
        int q = 0;
        for (unsigned long long m : {~0ull, 0x0101010101010101ull}) {
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kalt, dim3(blocks), dim3(64), 0, 0, out, cyc, m, 2048, K); hipDeviceSynchronize(); }
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            r[q++] = (double)h[blocks / 2] / 2048;
        }
        printf("alternating runs of %2d iterations: dense+dense %.1f cycles per pair of iterations, 8 lanes + dense %.1f -> sparse iteration %.1f, dense %.1f\n", K, r[0], r[1], r[1] - r[0] / 2, r[0] / 2);
    }
    for (int n : {8, 9}) {   // the lowest n lanes, and n lanes spread over the wave
        for (int spread = 0; spread < 2; ++spread) {
            unsigned long long m = 0;
            for (int i = 0; i < n; ++i) m |= 1ull << (spread ? (i * 64 / n) : i);
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, cyc, m, iters); hipDeviceSynchronize(); }
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("%2d lanes %-8s %.1f cycles per iteration\n", n, spread ? "spread" : "low", (double)h[blocks / 2] / iters);
        }
    }
    return 0;
}
