// rsx_math.hpp — device-side elementary functions with a FIXED operation order.
//
// The step engine promises results that are bit-identical to the scalar model in fp32, on any
// batch size / lane mapping.  libm-style device functions (ocml sinf/cosf/logf) carry no such
// promise, so the few transcendental functions the model needs are spelled out here as fixed
// mul / add / explicit-fma sequences (Cephes single-precision coefficients).  The translation
// unit is compiled with -ffp-contract=off: no multiply-add is ever fused behind the source's
// back (fused ones are written as fma_()); IEEE sqrt and divide are correctly rounded by hipcc's
// default expansion.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsx {

// clamp(v, lo, hi) for lo < hi, both non-zero: the median of three is exactly
// `v < lo ? lo : (v > hi ? hi : v)` for every non-NaN v (one v_med3_f32, no VCC round trip —
// on gfx950 a v_cmp -> v_cndmask pair costs two extra wait states).
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    return __builtin_amdgcn_fmed3f(v, lo, hi);
}
// -1 for negative x, +1 otherwise (only ever used where x != 0)
__device__ __forceinline__ float signf(float x) { return __builtin_copysignf(1.0f, x); }

// Lane <-> lane exchange through LDS inside ONE wavefront (every kernel here runs 64-thread
// workgroups).  The LDS unit executes a wave's DS instructions in issue order, so a ds_write is
// visible to a later ds_read of any lane of the same wave; all that is needed is that the compiler
// keeps the program order.  __syncthreads() would also do it, but its workgroup-scope fence
// drains vmcnt: every exchange would wait for the wave's outstanding global loads AND stores
// (a full HBM round trip per step in the fused kernels).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// fused multiply-add, written out: the translation unit is built with -ffp-contract=off, so an
// FMA exists exactly where the model says so (one rounding: v_fma_f32 == fmaf on the CPU side)
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// sin and cos of an angle already reduced to about [-pi, pi] (any |a| < 1e4 works).
__device__ __forceinline__ void sincos_f32(float a, float& s, float& c) {
    float t = a * 0.636619772f;                       // 2/pi
    int k = (int)(t + (t >= 0.0f ? 0.5f : -0.5f));    // nearest quadrant
    float fk = (float)k;
    // three-term Cody-Waite reduction by pi/2
    float r = fma_(fk, -7.54978995489188e-8f, fma_(fk, -4.837512969970703125e-4f, fma_(fk, -1.5703125f, a)));
    float z = r * r;
    float ps = fma_(fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    float pc = fma_(fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                    fma_(-0.5f, z, 1.0f));
    int q = k & 3;
    float ss = (q & 1) ? pc : ps;
    float cc = (q & 1) ? ps : pc;
    s = (q & 2) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}

// heading in degrees back into (-180, 180] after one sub-step's turn: "if (th > 180) th -= 360; else if (th < -180)
// th += 360" as two selects — the same values (both candidates are computed, one is picked), but no branch: the
// compiler turned the if / else-if into two save / restore sequences of the exec mask per robot and sub-step
__device__ __forceinline__ float wrap_deg(float th) {
    const float lo = th - 360.0f, hi = th + 360.0f;
    return th > 180.0f ? lo : (th < -180.0f ? hi : th);
}

// heading after a small turn d (rad): rotate (c, s) by sin / cos of d (|d| < 0.5; odd / even
// Taylor polynomials, error < 1e-8) — one exact sincos per step(), cheap rotations per sub-step
__device__ __forceinline__ void rotate_heading(float d, float& c, float& s) {
    float d2 = d * d;
    float sd = d * fma_(d2, fma_(d2, 8.3333333333333333e-3f, -1.6666666666666667e-1f), 1.0f);
    float cd = fma_(d2, fma_(d2, fma_(d2, -1.3888888888888889e-3f, 4.1666666666666664e-2f), -0.5f), 1.0f);
    float c0 = c, s0 = s;
    c = fma_(c0, cd, -(s0 * sd));
    s = fma_(s0, cd, c0 * sd);
}

// natural log for x in [2^-24, 1]
__device__ __forceinline__ float log_f32(float x) {
    uint32_t ix = __float_as_uint(x);
    int e = (int)(ix >> 23) - 127;
    float m = __uint_as_float((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }
    float f = m - 1.0f, z = f * f;
    float p = fma_(fma_(fma_(fma_(fma_(fma_(fma_(fma_(7.0376836292e-2f, f, -1.1514610310e-1f), f, 1.1676998740e-1f), f,
                    -1.2420140846e-1f), f, 1.4249322787e-1f), f, -1.6668057665e-1f), f, 2.0000714765e-1f), f,
                    -2.4999993993e-1f), f, 3.3333331174e-1f) * f * z;
    float fe = (float)e;
    p = fma_(fe, -2.12194440e-4f, p);
    p = fma_(-0.5f, z, p);
    float r = f + p;
    r = fma_(fe, 0.693359375f, r);
    return r;
}

// atan2 (Cephes atanf), fixed operation order — only the PassEndurance placement needs it
__device__ __forceinline__ float atan_f32(float x) {
    const float sgn = x < 0.0f ? -1.0f : 1.0f;
    x = fabsf(x);
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    const float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return sgn * y;
}
__device__ __forceinline__ float atan2_f32(float y, float x) {
    if (x > 0.0f) return atan_f32(y / x);
    if (x < 0.0f) return atan_f32(y / x) + (y >= 0.0f ? 3.14159265358979f : -3.14159265358979f);
    return y > 0.0f ? 1.5707963267948966f : (y < 0.0f ? -1.5707963267948966f : 0.0f);
}

// Philox4x32 (Salmon, Moraes, Dror, Shaw — SC'11).  counter = (global env id, episode,
// tick, domain), key = (seed lo, seed hi): a draw depends only on WHAT it is for, never on the
// thread that computes it, so results are invariant to batch size, batch position and sharding.
struct u32x4 { uint32_t x, y, z, w; };

// Rounds: 7, the smallest count of the family that passes BigCrush (SC'11, table 2) — 32-bit integer
// multiplies are quarter rate on CDNA, and at scale the draws were ~25 % of the step's VALU time
// with the 10-round default.  The round function is pinned by the published 10-round vectors.
constexpr int PHILOX_ROUNDS = 7;

__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// domains.  ACT: per-step block(s) of an env, block q in bits 8..: SSL tasks take the agent's action
// from block 0; VSS-v0 gives robot k the words (2 (k & 1), 2 (k & 1) + 1) of block k >> 1 (robot 0:
// random action, robots >= 1: the two uniforms of their Box-Muller OU draw) — one block serves two
// robots, which the one-lane-per-env kernel turns into half the Philox work.
constexpr uint32_t DOM_ACT = 1u, DOM_PLACE = 3u, DOM_RAW = 4u;   // RAW: rsx_step_dev_random, counter (env, tick, robot, RAW)

// 24-bit uniform in [0, 1)
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

}  // namespace rsx
