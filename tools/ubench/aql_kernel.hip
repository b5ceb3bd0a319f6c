// Device side of tools/ubench/aql_boundary.cpp: a stand-in for the headline step kernel (512 single-wave workgroups,
// 48 SoA rows of 8 envs read and written back, a straight-line dependent chain in between) whose dispatch packets are
// written by hand.   hipcc --offload-arch=gfx950 -O3 --cuda-device-only -c -o /tmp/aql_kernel.hsaco tools/ubench/aql_kernel.hip
#include <hip/hip_runtime.h>
#ifndef CHAIN
#define CHAIN 2600
#endif
extern "C" __global__ __launch_bounds__(64) void rows_chain(float* p, unsigned* xcc_bad, int B, int sc1_loads) {
    const int e = blockIdx.x * 8 + (threadIdx.x & 7), b = threadIdx.x >> 3;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0 && (xcc & 7u) != (blockIdx.x & 7u)) atomicAdd(xcc_bad, 1u);
    if (threadIdx.x == 0 && blockIdx.x < 16) xcc_bad[1 + blockIdx.x] = xcc;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { xcc_bad[32 + 2 * blockIdx.x] = hwid; xcc_bad[33 + 2 * blockIdx.x] = xcc; }   // where this wave ran
    if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned n = atomicAdd(&xcc_bad[1100], 1u); xcc_bad[1104 + (n & 63u)] = xcc & 15u; }   // block 0's XCC, launch by launch
    float v[6];
    if (sc1_loads) {
#pragma unroll
        for (int f = 0; f < 6; ++f) v[f] = __builtin_nontemporal_load(&p[(size_t)(b * 6 + f) * B + e]);
    } else {
#pragma unroll
        for (int f = 0; f < 6; ++f) v[f] = p[(size_t)(b * 6 + f) * B + e];
    }
    float x = ((v[0] + v[1]) + (v[2] + v[3])) + (v[4] + v[5]);
#pragma unroll
    for (int i = 0; i < CHAIN; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
#pragma unroll
    for (int f = 0; f < 6; ++f) p[(size_t)(b * 6 + f) * B + e] = v[f] + (x > 1e30f ? 1.0f : 0.0f) + (f == 0 ? 1.0f : 0.0f);
}
