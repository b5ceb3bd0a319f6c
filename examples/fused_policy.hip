// examples/fused_policy.hip — a 2-layer tanh MLP policy (obs -> 64 -> actions) as ONE hand-written gfx950 kernel, for the policy-in-the-loop
// example (examples/vec_policy_loop.py --fused-policy).  Not part of the engine: it stands where a trainer's inference code stands
// (the loop of the reference's README.md:116-133), to show what that loop costs when the policy is one launch instead of four small
// library kernels.  One wave = 64 lanes = the 64 hidden units; a wave serves ENVS_PER_WAVE envs: an env's observation row is fetched
// by one coalesced load (lane k holds obs[k]) and handed round with v_readlane, the hidden layer is 40 FMAs per lane against a weight
// column kept in registers, the output layer one lane per (env, action) summing its 64 products out of LDS.  No MFMA (10 MFLOP per
// 4096 envs).
#include <hip/hip_runtime.h>

namespace {
constexpr int H = 64;              // hidden units = lanes of a wave
constexpr int ENVS_PER_WAVE = 2;   // 4096 envs = 2048 waves: two per SIMD (a lone wave would wait out its own load latency)

__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); exact to ~2e-7 relative, saturates cleanly at +-1
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}

template <int OD, int A>
__global__ __launch_bounds__(64) void mlp_policy_kernel(const float* __restrict__ obs, const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ act,
                                                        const int B) {
    static_assert(ENVS_PER_WAVE * A <= 64, "one output lane per (env, action)");
    __shared__ float hs[ENVS_PER_WAVE][H];   // hidden activations of the wave's envs
    __shared__ float ws[A][H];               // second layer, one row per action
    const int j = threadIdx.x;
    float w[OD];
#pragma unroll
    for (int k = 0; k < OD; ++k) w[k] = w1[k * H + j];        // column j of the first layer: 64 contiguous floats per k across the wave
    const float bj = b1[j];
#pragma unroll
    for (int a = 0; a < A; ++a) ws[a][j] = w2[j * A + a];
    const int e0 = blockIdx.x * ENVS_PER_WAVE;
    float row[ENVS_PER_WAVE];
#pragma unroll
    for (int i = 0; i < ENVS_PER_WAVE; ++i) {                  // all observation rows of the wave in flight together with the weights
        const int e = e0 + i;
        row[i] = (e < B && j < OD) ? obs[(size_t)e * OD + j] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < ENVS_PER_WAVE; ++i) {
        float h = bj;
#pragma unroll
        for (int k = 0; k < OD; ++k) h = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(row[i]), k)), w[k], h);   // obs[k]: a scalar operand
        hs[i][j] = fast_tanh(h);
    }
    __syncthreads();
    // output layer: lane (i, a) sums its 64 products out of LDS (a cross-lane reduction per output would be 6 dependent exchanges each)
    if (j < ENVS_PER_WAVE * A) {
        const int i = j / A, a = j - i * A, e = e0 + i;
        float s0 = b2[a], s1 = 0.0f;
#pragma unroll
        for (int m = 0; m < H; m += 2) { s0 = fmaf(hs[i][m], ws[a][m], s0); s1 = fmaf(hs[i][m + 1], ws[a][m + 1], s1); }
        if (e < B) act[(size_t)e * A + a] = fast_tanh(s0 + s1);
    }
}
}  // namespace

// obs [B][od], w1 [od][64], b1 [64], w2 [64][a], b2 [a] -> act [B][a]; all device float32; stream = hipStream_t (capturable)
extern "C" int fused_mlp_policy(const float* obs, int B, int od, const float* w1, const float* b1, const float* w2, const float* b2,
                                float* act, int a, void* stream) {
    const dim3 grid((unsigned)((B + ENVS_PER_WAVE - 1) / ENVS_PER_WAVE)), block(64);
    hipStream_t s = (hipStream_t)stream;
    if (od == 40 && a == 2) hipLaunchKernelGGL((mlp_policy_kernel<40, 2>), grid, block, 0, s, obs, w1, b1, w2, b2, act, B);
    else if (od == 24 && a == 5) hipLaunchKernelGGL((mlp_policy_kernel<24, 5>), grid, block, 0, s, obs, w1, b1, w2, b2, act, B);
    else return -1;   // (the two shapes of the example: VSS-v0 and SSLStaticDefenders-v0)
    return (int)hipGetLastError();
}
