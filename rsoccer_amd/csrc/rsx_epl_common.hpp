// rsx_epl_common.hpp — what the one-lane-per-env kernels (rsx_epl.hpp: VSS-v0, rsx_epl_ssl.hpp: the four SSL tasks) share
// besides the per-body arithmetic of rsx_body.hpp: how a lane addresses its env's column of the SoA arrays, how the
// touching pairs of an env are enumerated and walked, where the contact sums live, how an observation row goes out.
#pragma once
#include "rsx_kernels.hpp"

namespace rsx {

// A lane's view of the [rows][B] arrays: buffer instructions — one resource per array in scalar registers, the row as the
// scalar offset, ONE 32-bit byte offset per lane (the env's column).  Plain pointer arithmetic compiled to a 64-bit
// vector add per row access (112 v_lshl_add_u64 + 60 v_mad_i64_i32 per VSS-v0 step) and kept row pointers alive in
// register pairs.  32-bit offsets: the host picks these kernels only while every array is smaller than 2 GB.
struct EplIO {
    __amdgpu_buffer_rsrc_t S, A;   // state, aux
    uint32_t eo;                   // 4 * env
    int B4;                        // bytes per row
    __device__ __forceinline__ EplIO(float* state, float* aux, int row_stride, int e)
        : S(__builtin_amdgcn_make_buffer_rsrc(state, 0, -1, 0x00020000)), A(__builtin_amdgcn_make_buffer_rsrc(aux, 0, -1, 0x00020000)),
          eo(4u * (uint32_t)e), B4(4 * row_stride) {}
    __device__ __forceinline__ float ld(const __amdgpu_buffer_rsrc_t rs, int row) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)eo, row * B4, 0));
    }
    __device__ __forceinline__ void st(const __amdgpu_buffer_rsrc_t rs, int row, float v) const {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)eo, row * B4, 0);
    }
    // six consecutive state rows of a robot (x, y, theta, vx, vy, omega: the wire format, Entities/Frame.py:20-47,55-92)
    __device__ __forceinline__ void st_robot(int row0, float x, float y, float th, float vx, float vy, float om) const {
        st(S, row0, x); st(S, row0 + 1, y); st(S, row0 + 2, th); st(S, row0 + 3, vx); st(S, row0 + 4, vy); st(S, row0 + 5, om);
    }
    __device__ __forceinline__ void st_flags(uint8_t* flags, int num_envs, int term, int trunc) const {
        const __amdgpu_buffer_rsrc_t FL = __builtin_amdgcn_make_buffer_rsrc(flags, 0, -1, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)term, FL, (int)(eo >> 2), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)trunc, FL, (int)(eo >> 2), num_envs, 0);
    }
};

// one observation row from registers ([B][OD] array, eo = 4 * env): 16-byte stores and a tail.  (Rows are OD * 4 bytes apart:
// with an odd OD the pieces are only 4-byte aligned — fine for buffer / global accesses on gfx9; twenty-one separate
// dword stores instead measured 20 % on the whole dribbling step.)
template <int OD>
__device__ __forceinline__ void epl_store_row(float* rows, const uint32_t eo, const float* ob) {
    const __amdgpu_buffer_rsrc_t O = __builtin_amdgcn_make_buffer_rsrc(rows, 0, -1, 0x00020000);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const int off = (int)(OD * eo);
#pragma unroll
    for (int i = 0; i < OD / 4; ++i) {
        const u4 v = {__builtin_bit_cast(unsigned, ob[4 * i]), __builtin_bit_cast(unsigned, ob[4 * i + 1]),
                      __builtin_bit_cast(unsigned, ob[4 * i + 2]), __builtin_bit_cast(unsigned, ob[4 * i + 3])};
        __builtin_amdgcn_raw_buffer_store_b128(v, O, off, 16 * i, 0);
    }
    constexpr int T = OD / 4 * 4;
    if (OD - T >= 2) {
        const u2 v = {__builtin_bit_cast(unsigned, ob[T]), __builtin_bit_cast(unsigned, ob[T + 1])};
        __builtin_amdgcn_raw_buffer_store_b64(v, O, off, 4 * T, 0);
    }
    if ((OD - T) & 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ob[OD - 1]), O, off, 4 * (OD - 1), 0);
}

// contact sums of a sub-step (only when some lane touches something), column = lane; NB bodies (the ball last)
template <int NB>
struct EplSums { float acc[4][NB][64]; float accw[64]; };
template <int NB>
__device__ __forceinline__ void epl_zero_sums(EplSums<NB>& c, const int lane) {
#pragma unroll
    for (int k = 0; k < NB; ++k) { c.acc[0][k][lane] = 0.0f; c.acc[1][k][lane] = 0.0f; c.acc[2][k][lane] = 0.0f; c.acc[3][k][lane] = 0.0f; }
    c.accw[lane] = 0.0f;
}

// pair p -> (i, j), i < j < NB, lexicographic (every body then receives its partners in index order): row r of the upper
// triangle starts at r NB - r (r + 1) / 2.  Sums of comparisons: nested ?: chains compiled to exec-mask branches.
template <int NB>
__device__ __forceinline__ void epl_pair(int p, int& i, int& j) {
    i = 0;
#pragma unroll
    for (int r = 1; r < NB - 1; ++r) i += (int)(p >= r * NB - r * (r + 1) / 2);
    const int start = i * NB - ((i * (i + 1)) >> 1);
    j = i + 1 + (p - start);
}
// bits of the pair set that involve body k
template <int NB>
__host__ __device__ constexpr unsigned epl_pair_mask(int k) {
    unsigned m = 0;
    int p = 0;
    for (int i = 0; i < NB; ++i)
        for (int j = i + 1; j < NB; ++j, ++p)
            if (i == k || j == k) m |= 1u << p;
    return m;
}
static_assert(epl_pair_mask<7>(0) == 0x00003Fu && epl_pair_mask<7>(3) == 0x038884u && epl_pair_mask<7>(6) == 0x1A4420u, "pair masks");

// Each lane walks ITS touching pairs in pair order.  Bodies 0..N-1 are the robots r[] (picked out of the registers by
// select chains: a snapshot in LDS would cost a wave of occupancy); BALL: body N is the ball as a circle (VSS).  Both sides
// of a pair from one normal (contact_pair, rsx_body.hpp); each body's sums are read-modify-written in LDS.
// ZCHK: the caller's pair bits come from `d2 < thr` alone (one compare per pair); the model's `0 < d2` is applied here, where a
// pair is walked: a pair of bodies in one place is dropped and reported in `dropped` (the caller clears its bit before the sums
// are applied).
template <int KIND, int N, bool BALL, bool ZCHK = false>
__device__ __forceinline__ void epl_walk_pairs(const Params& P, const Body* r, const Body& ball, EplSums<N + 1>& c, const int lane, unsigned todo, bool& deep, bool& wallp, unsigned* dropped = nullptr) {
    // does any robot of the wave's envs (with a touching pair) stand at a wall?  (wave-uniform, once per sweep: contact_pair, rsx_body.hpp)
    bool aw_ = false;
    if (KC<KIND>::wall_aware && todo) {
#pragma unroll
        for (int k = 0; k < N; ++k) aw_ |= at_wall<KIND>(P, r[k].x, r[k].y);
    }
    const bool v2w = KC<KIND>::wall_aware && __ballot(aw_) != 0ull;
    using K = KC<KIND>;
    while (todo) {
        const int p = __builtin_ctz(todo);
        todo &= todo - 1;
        int i, j;
        epl_pair<BALL ? N + 1 : N>(p, i, j);
        const bool rb = BALL && j == N;
        Body bi = Body{}, bj = Body{};
        float wi = 0.0f, wj = 0.0f;
#pragma unroll
        for (int k = 0; k < N; ++k) {   // selects, not branches
            { const bool m = i == k; bi.x = m ? r[k].x : bi.x; bi.y = m ? r[k].y : bi.y; bi.vx = m ? r[k].vx : bi.vx; bi.vy = m ? r[k].vy : bi.vy; wi = m ? r[k].om : wi; }
            { const bool m = j == k; bj.x = m ? r[k].x : bj.x; bj.y = m ? r[k].y : bj.y; bj.vx = m ? r[k].vx : bj.vx; bj.vy = m ? r[k].vy : bj.vy; wj = m ? r[k].om : wj; }
        }
        if (rb) { bj.x = ball.x; bj.y = ball.y; bj.vx = ball.vx; bj.vy = ball.vy; wj = ball.om; }
        if (ZCHK) {
            const float dx = bj.x - bi.x, dy = bj.y - bi.y;
            if (__builtin_expect(fma_(dx, dx, dy * dy) == 0.0f, 0)) { *dropped |= 1u << p; continue; }
        }
        const float lever_j = rb ? K::r_ball : K::r_robot;
        float ai[4] = {c.acc[0][i][lane], c.acc[1][i][lane], c.acc[2][i][lane], c.acc[3][i][lane]};
        float aj[4] = {c.acc[0][j][lane], c.acc[1][j][lane], c.acc[2][j][lane], c.acc[3][j][lane]};
        float awj = rb ? c.accw[lane] : 0.0f;
        contact_pair<KIND>(P, bi, bj, fma_(wj, lever_j, wi * K::r_robot), fma_(wi, K::r_robot, wj * lever_j), rb ? K::rs_rb : K::rs_rr,
                           rb ? K::ope_rb : K::ope_rr, rb ? K::w_rb_r : K::w_rr, rb ? K::w_rb_b : K::w_rr, rb ? K::kt_rb_r : K::kt_rr,
                           rb ? K::kt_rb_b : K::kt_rr, rb ? K::mu_rb : K::mu_rr, rb ? K::spin_c : 0.0f, K::beta, K::pen2, !rb, v2w, ai, aj, awj, deep, wallp);
        c.acc[0][i][lane] = ai[0]; c.acc[1][i][lane] = ai[1]; c.acc[2][i][lane] = ai[2]; c.acc[3][i][lane] = ai[3];
        c.acc[0][j][lane] = aj[0]; c.acc[1][j][lane] = aj[1]; c.acc[2][j][lane] = aj[2]; c.acc[3][j][lane] = aj[3];
        if (rb) c.accw[lane] = awj;
    }
}

// only a body that touched something is updated (the others keep their bits); touched(k): body k (N = the ball) has a sum
template <int N, typename F>
__device__ __forceinline__ void epl_apply_sums(Body* r, Body& ball, const EplSums<N + 1>& c, const int lane, F touched) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (touched(k)) {
            r[k].vx = r[k].vx + c.acc[0][k][lane]; r[k].vy = r[k].vy + c.acc[1][k][lane];
            r[k].x = r[k].x + c.acc[2][k][lane]; r[k].y = r[k].y + c.acc[3][k][lane];
        }
    }
    if (touched(N)) {
        ball.vx = ball.vx + c.acc[0][N][lane]; ball.vy = ball.vy + c.acc[1][N][lane];
        ball.x = ball.x + c.acc[2][N][lane]; ball.y = ball.y + c.acc[3][N][lane];
        ball.om = ball.om + c.accw[lane];
    }
}

}  // namespace rsx
