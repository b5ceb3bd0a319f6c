// valu_rates.hip — issue cost of the VALU instructions the step kernels are made of, on gfx950.
// For W = 1..4 waves per SIMD: shader cycles (s_memtime) one wave needs per instruction when every
// SIMD of the chip holds W such waves.  cycles x 1/W = the SIMD's issue cost of the instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
constexpr int ITERS = 2000;

#define KERNEL(name, body)                                                                     \
__global__ __launch_bounds__(64) void name(float* out, unsigned long long* cyc) {              \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f; \
    const float k = 1.0000001f, m = 0.9999999f;                                                \
    unsigned long long t0 = __builtin_readcyclecounter();                                      \
    for (int i = 0; i < ITERS; ++i) { body }                                                   \
    unsigned long long t1 = __builtin_readcyclecounter();                                      \
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7; \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                           \
}

// 16 independent instructions per iteration (8 chains x 2)
#define OP1(ins) \
    asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" \
                 ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); \
    asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" \
                 ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n" \
                 : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(m));
#define OP3(ins) \
    asm volatile(ins " %0, %0, %8, %0\n" ins " %1, %1, %8, %1\n" ins " %2, %2, %8, %2\n" ins " %3, %3, %8, %3\n" \
                 ins " %4, %4, %8, %4\n" ins " %5, %5, %8, %5\n" ins " %6, %6, %8, %6\n" ins " %7, %7, %8, %7\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); \
    asm volatile(ins " %0, %0, %8, %0\n" ins " %1, %1, %8, %1\n" ins " %2, %2, %8, %2\n" ins " %3, %3, %8, %3\n" \
                 ins " %4, %4, %8, %4\n" ins " %5, %5, %8, %5\n" ins " %6, %6, %8, %6\n" ins " %7, %7, %8, %7\n" \
                 : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(m));
#define OPU(ins) \
    asm volatile(ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3\n" \
                 ins " %4, %4\n" ins " %5, %5\n" ins " %6, %6\n" ins " %7, %7\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); \
    asm volatile(ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3\n" \
                 ins " %4, %4\n" ins " %5, %5\n" ins " %6, %6\n" ins " %7, %7\n" \
                 : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
// packed: 8 instructions on register pairs (a0:a1 must be consecutive -> use 64-bit operands)
typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(name, body)                                                                    \
__global__ __launch_bounds__(64) void name(float* out, unsigned long long* cyc) {              \
    f2 a0 = {threadIdx.x + 0.f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
    f2 b0 = a0 * .5f, b1 = a1 * .5f, b2 = a2 * .5f, b3 = a3 * .5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f; \
    const f2 k = {1.0000001f, 0.9999999f}, m = {0.9999999f, 1.0000001f};                       \
    unsigned long long t0 = __builtin_readcyclecounter();                                      \
    for (int i = 0; i < ITERS; ++i) { body }                                                   \
    unsigned long long t1 = __builtin_readcyclecounter();                                      \
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;     \
    out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;                                            \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                           \
}

KERNEL(k_fma, OP3("v_fma_f32"))
KERNEL(k_mul, OP1("v_mul_f32"))
KERNEL(k_add, OP1("v_add_f32"))
KERNEL(k_med3, OP3("v_med3_f32"))
KERNEL(k_mullo, OP1("v_mul_lo_u32"))
KERNEL(k_mulhi, OP1("v_mul_hi_u32"))
KERNEL(k_xor, OP1("v_xor_b32"))
KERNEL(k_sqrt, OPU("v_sqrt_f32"))
KERNEL(k_rcp, OPU("v_rcp_f32"))
KERNEL(k_cvt, OPU("v_cvt_f32_u32"))
KERNEL(k_mov, OPU("v_mov_b32"))
KERNEL(k_max, OP1("v_max_f32"))
KERNEL(k_cndmask, OP1("v_cndmask_b32"))
KERNEL2(k_pk_fma, OP3("v_pk_fma_f32"))
KERNEL2(k_pk_mul, OP1("v_pk_mul_f32"))
KERNEL2(k_pk_add, OP1("v_pk_add_f32"))
// dependent chain: latency
__global__ __launch_bounds__(64) void k_fma_dep(float* out, unsigned long long* cyc) {
    float a0 = threadIdx.x; const float k = 1.0000001f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile(REP8("v_fma_f32 %0, %0, %1, %0\n") REP8("v_fma_f32 %0, %0, %1, %0\n") : "+v"(a0) : "v"(k));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// v_cmp (writes an SGPR pair) + v_cndmask reading it: the pair the clamps avoid
__global__ __launch_bounds__(64) void k_cmp_cnd(float* out, unsigned long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; const float k = 17.0f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %1, vcc\n") REP8("v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %3, vcc\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int per_iter) {
    float* out; unsigned long long* cyc;
    const int maxb = 1024 * 8;
    hipMalloc(&out, maxb * 64 * sizeof(float)); hipMalloc(&cyc, maxb * sizeof(unsigned long long));
    printf("%-12s", name);
    for (int W : {1, 2, 3, 4, 8}) {
        const int blocks = 1024 * W;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[blocks / 2] / ((double)ITERS * per_iter);
        // wall: instructions per SIMD = W * ITERS * per_iter; ns per instruction per SIMD
        const double ns = ms * 1e6 / ((double)W * ITERS * per_iter);
        printf("  W=%d: %6.2f cyc/instr/wave (%5.2f /W) %6.3f ns/instr/SIMD |", W, med, med / W, ns);
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run("v_fma_f32", k_fma, 16); run("v_mul_f32", k_mul, 16); run("v_add_f32", k_add, 16);
    run("v_med3_f32", k_med3, 16); run("v_max_f32", k_max, 16); run("v_cndmask", k_cndmask, 16);
    run("v_mov_b32", k_mov, 16); run("v_xor_b32", k_xor, 16);
    run("v_pk_fma_f32", k_pk_fma, 16); run("v_pk_mul_f32", k_pk_mul, 16); run("v_pk_add_f32", k_pk_add, 16);
    run("v_mul_lo_u32", k_mullo, 16); run("v_mul_hi_u32", k_mulhi, 16);
    run("v_sqrt_f32", k_sqrt, 16); run("v_rcp_f32", k_rcp, 16); run("v_cvt_f32_u32", k_cvt, 16);
    run("fma dep", k_fma_dep, 16); run("cmp+cndmask", k_cmp_cnd, 32);
    return 0;
}
