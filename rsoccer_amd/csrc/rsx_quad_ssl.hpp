// rsx_quad_ssl.hpp — the SSL 11v11 scrimmage task (BASELINE.json configs[3]: 22 robots + ball on the division-A
// field, every robot commanded on the device) for LARGE batches: "four lanes per env, six robots per lane".
//
// Same path, same buffers, same arithmetic as task_step_kernel<SSL, 32, SCRIMMAGE, 22, ...> (rsx_kernels.hpp; the
// reference's hot path: robosim.SSL.step / get_state, rsoccer_gym/Simulators/rsim.py:128-158, under a README.md:78-110
// style task) — other mapping: a DPP quad (lanes 4q .. 4q+3) owns env q of a 16-env tile; lane p holds the robots
// 6p .. 6p+5 in registers (lane 3: robots 18..21, its last two slots are ghosts at NaN positions that can touch
// nothing, and the ball).  What it buys over 32 lanes per env (one body per lane, 9 lanes idle, every pair tested
// from both sides through LDS): all lanes busy with robots, the per-step work (Philox block, targets, wheel speeds,
// eleven row stores per robot) issued once for sixteen envs instead of two, every robot pair of an env tested ONCE and on
// registers — the other lanes' positions are DPP quad_perm reads, the result is one bit per pair that both robots' lanes
// get (no LDS on the path without contacts).  What it costs:
// a lane walks six robots one after the other, so it needs enough envs to fill the chip (the host picks it from
// RSX_QUAD_MIN_ENVS on).  Results are bit-identical: per-body formulas are rsx_body.hpp's, every body sums its
// partners in index order (robots, then the ball), the ball sums the robots' records in robot order, draws use the same
// Philox counters (tests/test_gpu_parity.py::test_quad_layout_is_bit_identical).
#pragma once
#include "rsx_kernels.hpp"

#ifndef RSX_QUAD_WAVES
#define RSX_QUAD_WAVES 3   // waves per SIMD the kernel is compiled for
#endif

namespace rsx {

#ifdef RSX_QSTATS   // development: what the contact path of a launch did (tools/exp_quad_stats.py)
__device__ unsigned long long rsx_qstats[16];
#define RSX_QS(I, V) do { if (lane == 0) atomicAdd(&rsx_qstats[I], (unsigned long long)(V)); } while (0)
__device__ __forceinline__ unsigned qs_wave_max(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ unsigned qs_wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = v + (unsigned)__shfl_xor((int)v, o);
    return v;
}
#else
#define RSX_QS(I, V) do {} while (0)
#endif

constexpr int Q_N = 22;        // robots
constexpr int Q_R = 6;         // robot slots per lane
constexpr int Q_ENVS = 16;     // envs per wave
constexpr int Q_OD = 2 + 2 * Q_N;   // observation width of the scrimmage task

struct QuadShared {
    // snapshot of a contact sweep, slot = robot index, column = env of the wave: A = (x, y, vx, vy), W = yaw rate, CS = (cos, sin)
    // of the heading; read by index on the contact path only.  9856 bytes: sixteen waves of it fit a CU's 160 KB.
    float4 A[Q_N][Q_ENVS];
    float W[Q_N][Q_ENVS];
    float2 CS[Q_N][Q_ENVS];
};
static_assert(sizeof(QuadShared) <= 10240, "four waves per SIMD");

template <int CTRL>
__device__ __forceinline__ float qdpp_f(float v) {   // the same register of another lane of the quad (a DPP operand)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned qdpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int Q_NEXT = 0x39;   // quad_perm:[1,2,3,0]: lane p reads lane p + 1
constexpr int Q_DIAG = 0x4E;   // quad_perm:[2,3,0,1]: lane p reads lane p + 2
constexpr int Q_PREV = 0x93;   // quad_perm:[3,0,1,2]: lane p reads lane p - 1
constexpr int Q_L3 = 0xFF;     // quad_perm:[3,3,3,3]: the ball lane's value in all four lanes

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RSX_QUAD_WAVES, RSX_QUAD_WAVES))) void ssl_quad_kernel(RSX_HOT_ARGS, const Params P_, const Buffers bufs_) {
    static_assert(MODE == MODE_STEP, "single-step launches");
    constexpr int KIND = RSX_KIND_SSL, TASK = RSX_TASK_SSL_SCRIMMAGE, N = Q_N, R = Q_R, RS = 11;
    using K = KC<KIND>;
    using T = TC<TASK>;
    Params P = P_; RSX_UNPACK_HOT(P);
    Buffers bufs = bufs_; bufs.state = hp_state; bufs.aux = hp_aux; bufs.actions = hp_in; bufs.flags = hp_flags;
    __shared__ QuadShared sh;
    const int lane = threadIdx.x;
    const int q = lane >> 2;           // env of the wave
    const int p = lane & 3;            // which quarter of the robots
    const bool bl = p == 3;            // the ball's lane (reward, termination, episode bookkeeping of the env)
    const bool tick_dev = (hp_n_steps & RSX_TICK_DEV) != 0;   // step counter of this launch: rsx_kernels.hpp, step_tick
    const StepTick tk = step_tick(tick_dev, P, bufs, 1u);
    if (__builtin_expect(!tk.ok, 0)) return;
    const int tile = tile_of_block_zigzag(zigzag_per(tick_dev, tk.t, hp_per_xcd));
    const int e_raw = tile * Q_ENVS + q;
    int live_i = e_raw < P.num_envs ? 1 : 0;
    asm volatile("" : "+v"(live_i));               // decided here: one flag through the step, not the index it is made of
    const bool live = live_i != 0;
    const int e = live ? e_raw : P.num_envs - 1;   // lanes beyond the batch shadow its last env (loads valid, stores masked)
    const uint32_t env_id = P.env_id_base + (uint32_t)e;
    // buffer addressing (rsx_epl.hpp): resource per array, row as the scalar offset, one 32-bit lane offset per array
    const uint32_t eo = 4u * (uint32_t)e;                                              // the env's column
    const uint32_t ro = eo + 4u * (uint32_t)(5 + RS * R * p) * (uint32_t)P.row_stride;     // ... from this lane's first robot row
    const __amdgpu_buffer_rsrc_t S = __builtin_amdgcn_make_buffer_rsrc(bufs.state, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t A = __builtin_amdgcn_make_buffer_rsrc(bufs.aux, 0, -1, 0x00020000);
    const int B4 = 4 * P.row_stride;
    auto ld = [](const __amdgpu_buffer_rsrc_t rs, int row_off, uint32_t off) -> float {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, row_off, 0));
    };
    auto stf = [](const __amdgpu_buffer_rsrc_t rs, int row_off, uint32_t off, float v) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)off, row_off, 0);
    };
    // slot m of this lane is robot 6p + m; the last two slots of lane 3 are ghosts
    auto real = [&](int m) -> bool { return m < N - R * 3 || !bl; };
    const float qnan = __int_as_float(0x7FC00000);

    // ---- load ----
    float raw[R][6], rawb[7] = {0, 0, 0, 0, 0, 0, 0};
    float info[2] = {0, 0}, ep_ret = 0.0f;
    int steps = 0; uint32_t episode = 0;
    const bool fed = bufs.actions != nullptr;
#pragma unroll
    for (int m = 0; m < R; ++m) {
        // (ghost slots read the rows of robots 16, 17 — valid memory — and are overwritten below)
        const uint32_t rom = (m >= N - R * 3 && bl) ? ro - 4u * (uint32_t)(RS * 2) * (uint32_t)P.row_stride : ro;
#pragma unroll
        for (int f = 0; f < 6; ++f) raw[m][f] = ld(S, (RS * m + f) * B4, rom);
    }
    if (bl) {
#pragma unroll
        for (int f = 0; f < 5; ++f) rawb[f] = ld(S, f * B4, eo);
        rawb[5] = ld(S, P.state_dim * B4, eo);
        rawb[6] = ld(S, (P.state_dim + 1) * B4, eo);
    }
    float4 act[R];
#pragma unroll
    for (int m = 0; m < R; ++m) act[m] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (fed) {   // [B][N][4]: sixteen bytes per robot
        const __amdgpu_buffer_rsrc_t AC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bufs.actions), 0, -1, 0x00020000);
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int k = real(m) ? R * p + m : 0;
            const u4 v = __builtin_amdgcn_raw_buffer_load_b128(AC, (int)((uint32_t)(e * N + k) * 16u), 0, 0);
            act[m] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));   // (indexed: the .x/.y/.z/.w accessors of this vector type narrow the load to one dword on this compiler)
        }
    }
    const bool counts_steps = blockIdx.x == 0 && lane == 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

    Body r[R];
    Body ball = Body{};
#pragma unroll
    for (int m = 0; m < R; ++m) {   // interpret_body, robot
        r[m] = Body{};
        const bool rl = real(m);
        r[m].x = rl ? raw[m][0] : qnan; r[m].y = rl ? raw[m][1] : qnan; r[m].vx = raw[m][3]; r[m].vy = raw[m][4];
        r[m].th = raw[m][2];
        r[m].om = raw[m][5] * K::deg2rad;
        sincos_f32(r[m].th * K::deg2rad, r[m].s, r[m].c);
        __builtin_amdgcn_sched_barrier(0);
    }
    ball.x = rawb[0]; ball.y = rawb[1]; ball.vx = rawb[3]; ball.vy = rawb[4];
    ball.z = bl ? rawb[2] - K::r_ball : 0.0f; ball.vz = rawb[5]; ball.om = rawb[6];   // (the other lanes carry an all-zero ball)

    const uint32_t t = tk.t;
    unsigned kickbits = 0;
    // ---- actions -> commands: every robot (v_x, v_y, v_theta, kick), block 6p + m of the step ----
#pragma unroll
    for (int m = 0; m < R; ++m) {
        const uint32_t b = (uint32_t)(R * p + m);
        float a[4];
        if (fed) { a[0] = act[m].x; a[1] = act[m].y; a[2] = act[m].z; a[3] = act[m].w; }
        else {
            const u32x4 u = philox4x32(env_id, 0u, t, DOM_ACT | (b << 8), P.key0, P.key1);
            a[0] = u01(u.x) * 2.0f - 1.0f; a[1] = u01(u.y) * 2.0f - 1.0f; a[2] = u01(u.z) * 2.0f - 1.0f; a[3] = u01(u.w) * 2.0f - 1.0f;
        }
        float qc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        qc[1] = a[0] * T::max_v; qc[2] = a[1] * T::max_v; qc[3] = a[2] * 10.0f;
        qc[5] = a[3] > 0.9f ? 5.0f : 0.0f;
        robot_targets<KIND>(P, r[m], qc);
        kickbits |= a[3] > 0.9f ? 1u << m : 0u;   // the kick command (5 m/s or nothing): one bit, not a register, through the physics
        r[m].kick_x = 0.0f;
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- physics ----
    ball_step_friction(P, ball);
    unsigned irbits = 0;   // infrared flags of this lane's robots (bit m), refreshed by the first sweep of every sub-step
    for (int sub = 0; sub < P.n_sub; ++sub) {
        // A: actuation + integration
#pragma unroll
        for (int m = 0; m < R; ++m) { integrate_robot<KIND>(P, r[m]); __builtin_amdgcn_sched_barrier(0); }
        integrate_ball<KIND>(P, ball);

        // B: contacts.  First the pair test — (my robots x my robots), (mine x the next lane's), (mine x half of the lane after
        // that's): together the four lanes cover every robot pair of the env once — and (mine x ball); no LDS while nothing touches.
        const bool ball_low = qdpp_f<Q_L3>(ball.z) < K::robot_h;
        constexpr float NEAR2 = 0.13f * 0.13f;   // > (dck_rb + ir_tol)^2 + half_kw^2 = 0.126^2 and > rs_rb^2 (rsx_epl_ssl.hpp)
        irbits = 0;
        BallOverride bo{false, false, 0.0f, 0.0f, 0.0f};
        bool deep_env = false, wall_env = false;
        RSX_QS(0, 1);
        for (int sweep = 0; sweep < 4; ++sweep) {
            // second sweep: envs with a deep pair only; third and fourth: envs whose last sweep also saw a wall pair (model v2)
            const bool active = sweep == 0 || (deep_env && (sweep == 1 || wall_env));
            if (sweep >= 1 && !__any(active)) break;   // no env of the wave goes on: no further pair test either
            auto dist2 = [](float xj, float yj, float xi, float yi) -> float {
                const float dx = xj - xi, dy = yj - yi;
                return fma_(dx, dx, dy * dy);
            };
            // the env's ball in all four lanes (DPP reads need the source lane ACTIVE: all cross-lane reads of a sweep
            // happen here and at its other wave-uniform points, never inside a lane-divergent branch)
            const float bx = qdpp_f<Q_L3>(ball.x), by = qdpp_f<Q_L3>(ball.y);
            const float bvx = qdpp_f<Q_L3>(ball.vx), bvy = qdpp_f<Q_L3>(ball.vy), bom = qdpp_f<Q_L3>(ball.om);
            unsigned tf = 0, nf = 0;   // bit i: my robot i is closer than two radii to something / is near the ball
            // Every robot pair of the env ONCE, as one bit: d2 < (2 r)^2 (d2 is the same float from either side: (xj - xi)^2 =
            // (xi - xj)^2; NaN — a ghost slot — compares false; a pair at distance exactly 0, which the model does not count as
            // touching, is dropped where the partners are walked).  A lane tests its robots against the next lane's (36 pairs),
            // against the lane after next's where i <= j (21: the other half is computed over there) and against each other (15).
            // A compare feeds TWO shift registers (acc = 2 acc + bit: the carry-in of an add): the partner mask of my robot and the
            // mask of the OTHER lane's robot over mine, which goes there packed six bits per robot through two DPP reads.
            // rel[i]: bits 0..5 my own robots, 6..11 the next lane's, 12..17 the lane after next's, 18..23 the previous lane's
#ifdef RSX_QSTATS
            if (sub == 0 && sweep == 0) {   // how often is every pair of the wave far enough apart that no contact can occur in this step?
                float mn = 1.0e30f;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const float xn = qdpp_f<Q_NEXT>(r[j].x), yn = qdpp_f<Q_NEXT>(r[j].y), xd = qdpp_f<Q_DIAG>(r[j].x), yd = qdpp_f<Q_DIAG>(r[j].y);
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const float a = dist2(xn, yn, r[i].x, r[i].y), b = dist2(xd, yd, r[i].x, r[i].y);
                        if (a < mn) mn = a;
                        if (b < mn) mn = b;
                        if (i < j) { const float c = dist2(r[j].x, r[j].y, r[i].x, r[i].y); if (c < mn) mn = c; }
                    }
                }
                const float wmn = __uint_as_float(~qs_wave_max(~__float_as_uint(mn)));   // min of non-negative floats
                RSX_QS(13, wmn > 0.26f * 0.26f); RSX_QS(14, wmn > 0.30f * 0.30f); RSX_QS(15, wmn > 0.34f * 0.34f);
            }
#endif
            unsigned rel[R];
            {
                auto key = [](float xj, float yj, float xi, float yi) -> float {
                    const float dx = xj - xi, dy = yj - yi;
                    return fma_(dx, dx, dy * dy);
                };
                unsigned long long cdump;   // the (always clear) carry-out of the first add
#define RSX_QPUSH2(A, B, U) asm("v_cmp_gt_f32_e32 vcc, %4, %3\n\tv_addc_co_u32_e64 %0, %2, %0, %0, vcc\n\tv_addc_co_u32_e32 %1, vcc, %1, %1, vcc" \
                                : "+v"(A), "+v"(B), "=&s"(cdump) : "v"(U), "s"(K::rs_rr2) : "vcc")
#pragma unroll
                for (int i = 0; i < R; ++i) rel[i] = 0u;
                unsigned cw0 = 0u, cw1 = 0u;   // the next lane's robots over mine: robot j in bits 6 j .. 6 j + 5 of cw0 (j < 5), robot 5 in cw1
#pragma unroll
                for (int j = R - 1; j >= 0; --j) {   // highest first: the bits are shifted in from the right
                    const float xn = qdpp_f<Q_NEXT>(r[j].x), yn = qdpp_f<Q_NEXT>(r[j].y);
                    unsigned cf = 0u;
#pragma unroll
                    for (int i = R - 1; i >= 0; --i) { const float u = key(xn, yn, r[i].x, r[i].y); RSX_QPUSH2(rel[i], cf, u); }
                    if (j == R - 1) cw1 = cf; else cw0 = (cw0 << 6) | cf;
                    nf = (nf + nf) + (unsigned)(ball_low & (dist2(bx, by, r[j].x, r[j].y) < NEAR2));
                    __builtin_amdgcn_sched_barrier(0);
                }
                {   // what the previous lane found about my robots -> bits 18..23; mine x next -> bits 6..11
                    const unsigned pw0 = qdpp_u<Q_PREV>(cw0), pw1 = qdpp_u<Q_PREV>(cw1);
#pragma unroll
                    for (int i = 0; i < R; ++i) rel[i] = (rel[i] << 6) | ((i == R - 1 ? pw1 : ((pw0 >> (6 * i)) & 63u)) << 18);
                }
                unsigned dw0 = 0u, dw1 = 0u;
                unsigned dg[R];
#pragma unroll
                for (int i = 0; i < R; ++i) dg[i] = 0u;
#pragma unroll
                for (int j = R - 1; j >= 0; --j) {
                    const float xd = qdpp_f<Q_DIAG>(r[j].x), yd = qdpp_f<Q_DIAG>(r[j].y);
                    unsigned cd = 0u;
#pragma unroll
                    for (int i = j; i >= 0; --i) { const float u = key(xd, yd, r[i].x, r[i].y); RSX_QPUSH2(dg[i], cd, u); }
                    if (j == R - 1) dw1 = cd; else dw0 = (dw0 << 6) | cd;
                    __builtin_amdgcn_sched_barrier(0);
                }
                {   // my robot i: its partners j >= i over there from my tests (pushed for j = 5 .. i: bit j sits at j - i), j <= i from theirs
                    const unsigned ew0 = qdpp_u<Q_DIAG>(dw0), ew1 = qdpp_u<Q_DIAG>(dw1);
#pragma unroll
                    for (int i = 0; i < R; ++i) rel[i] |= ((dg[i] << i) | (i == R - 1 ? ew1 : ((ew0 >> (6 * i)) & 63u))) << 12;
                }
                unsigned own[R];
#pragma unroll
                for (int i = 0; i < R; ++i) own[i] = 0u;
#pragma unroll
                for (int j = R - 1; j >= 1; --j) {   // robot a receives its partners 5 .. a + 1 as the lower one of a pair, a zero for itself, then a - 1 .. 0
                    own[j] = own[j] + own[j];
#pragma unroll
                    for (int i = j - 1; i >= 0; --i) { const float u = key(r[j].x, r[j].y, r[i].x, r[i].y); RSX_QPUSH2(own[j], own[i], u); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                own[0] = own[0] + own[0];
#pragma unroll
                for (int i = 0; i < R; ++i) rel[i] |= own[i];
#undef RSX_QPUSH2
            }
#pragma unroll
            for (int i = R - 1; i >= 0; --i) tf = (tf + tf) + (unsigned)(rel[i] != 0u);
            if (!active) { tf = 0; nf = 0; }
            if (!__any((tf | nf) != 0)) break;
            RSX_QS(1, 1); if (sweep == 1) RSX_QS(11, 1);

            // ---- some env of the wave has a contact: snapshot -> LDS; then robot slot by robot slot (only the slots that
            // are flagged in some env of the wave): walk, the robot's side of its ball contact ----
#pragma unroll
            for (int m = 0; m < R; ++m) {
                if (real(m)) {
                    sh.A[R * p + m][q] = make_float4(r[m].x, r[m].y, r[m].vx, r[m].vy);
                    sh.W[R * p + m][q] = r[m].om;
                    sh.CS[R * p + m][q] = make_float2(r[m].c, r[m].s);
                }
            }
            wave_sync();
            const unsigned near = nf;
            bool deep = false, wallp = false;
            bool aw_ = false;   // does any robot of the wave's active envs stand at a wall?  (wave-uniform, once per sweep: contact_response, rsx_body.hpp)
            if (active) {
#pragma unroll
                for (int m = 0; m < R; ++m) aw_ |= at_wall<KIND>(P, r[m].x, r[m].y);   // (a ghost slot's NaN compares false)
            }
            const bool v2w = __ballot(aw_) != 0ull;
            unsigned rb_touch = 0;   // my robots that touch the ball in this sweep
            unsigned todo[R];   // partner sets (Jacobi: nothing has moved since the pair test)
            {   // bit = robot index: the relative order rotated into place by 6 p
                const unsigned sh6 = 6u * (unsigned)p;
#pragma unroll
                for (int i = 0; i < R; ++i) todo[i] = active ? (((rel[i] << sh6) | (rel[i] >> (24u - sh6))) & 0xFFFFFFu) : 0u;
            }
#ifdef RSX_QSTATS
            {
                unsigned tot = 0, trips = 0;
#pragma unroll
                for (int i = 0; i < R; ++i) { const unsigned c = (unsigned)__builtin_popcount(todo[i]); tot += c; trips += qs_wave_max(c); }
                const unsigned flat = qs_wave_max(tot), all = qs_wave_sum(tot);
                RSX_QS(5, trips); RSX_QS(6, all); RSX_QS(9, flat);
                const unsigned nn = qs_wave_sum((unsigned)__builtin_popcount(near));
                RSX_QS(12, nn);
            }
#endif
#pragma unroll
            for (int i = 0; i < R; ++i) {   // each robot of the lane: its robot partners in index order, then the ball
                if (__any(((tf | near) >> i) & 1u)) {
                    RSX_QS(4, 1);
                    const unsigned todo_i = todo[i];
                    float avx = 0.0f, avy = 0.0f, apx = 0.0f, apy = 0.0f, unused = 0.0f;
                    unsigned td = todo_i;
                    bool hit = false;
                    while (td) {
                        const int j = __builtin_ctz(td);
                        td &= td - 1;
                        const float4 oj = sh.A[j][q];
                        const float wj = sh.W[j][q];
                        const float dx = oj.x - r[i].x, dy = oj.y - r[i].y;
                        const float d2 = fma_(dx, dx, dy * dy);
                        if (d2 > 0.0f) {   // (0: two robots in one place are not a contact, rsx_kernels.hpp)
                            hit = true;
                            contact_response<KIND>(P, r[i], oj, d2, K::rs_rr, K::ope_rr, K::w_rr, K::kt_rr, K::mu_rr, 0.0f,
                                                   fma_(wj, K::r_robot, r[i].om * K::r_robot), K::beta, K::pen2, true, v2w, avx, avy, apx, apy, unused, deep, wallp);
                        }
                    }
                    bool touch = false;
                    if ((near >> i) & 1u) {   // the robot's side of its robot-ball contact (kicker mouth or body circle), infrared
                        const Body& o = r[i];
                        float dx = bx - o.x, dy = by - o.y;
                        float nx = 0.0f, ny = 0.0f, pen = -1.0f;
                        bool mouth = false;
                        float lx = fma_(dx, o.c, dy * o.s), ly = fma_(dy, o.c, -(dx * o.s));
                        if (fabsf(ly) < K::half_kw && lx > 0.0f) {
                            mouth = true; pen = K::dck_rb - lx; nx = o.c; ny = o.s; touch = pen > 0.0f;
                        } else {
                            float d2 = fma_(dx, dx, dy * dy);
                            if (d2 < K::rs_rb2 && d2 > 0.0f) {
                                float d = sqrtf(d2), inv = 1.0f / d;
                                nx = dx * inv; ny = dy * inv; pen = K::rs_rb - d; touch = true;
                            }
                        }
                        if (touch) {
                            deep |= pen > K::pen2;
                            const float dvx = bvx - o.vx, dvy = bvy - o.vy;
                            float vn = fma_(dvx, nx, dvy * ny);
                            if (vn < 0.0f) {
                                float qq = K::ope_rb * vn * K::w_rb_r; avx = fma_(qq, nx, avx); avy = fma_(qq, ny, avy);
                                const float wsum = fma_(bom, K::r_ball, o.om * (mouth ? K::dck : K::r_robot));
                                const float vt = fma_(dvy, nx, -(dvx * ny)) - wsum;
                                const float lim = qq * K::mu_rb;
                                const float ft = clampf(vt * K::kt_rb_r, lim, -lim);
                                avx = fma_(-ft, ny, avx); avy = fma_(ft, nx, avy);
                            }
                            float pc = K::beta * pen * K::w_rb_r;
                            apx = fma_(-pc, nx, apx); apy = fma_(-pc, ny, apy);
                            rb_touch |= 1u << i;
                        }
                        if (sweep == 0 && mouth && pen > -K::ir_tol) irbits |= 1u << i;
                    }
                    if (hit | touch) {   // only a body that touched something is updated
                        r[i].vx = r[i].vx + avx; r[i].vy = r[i].vy + avy;
                        r[i].x = r[i].x + apx; r[i].y = r[i].y + apy;
                    }
                }
            }
            // the ball's side: the robots that touch it (or hold it in front of a kicker that fires), in robot order, from the snapshot
            // — the same expressions the robot's lane evaluated (its own view of the contact), so the same numbers and the same
            // decisions: a robot whose lane found neither a touch nor a kick adds nothing here either; kicker: the last robot in
            // index order wins
            const unsigned mine = rb_touch | (sweep == 0 ? (irbits & kickbits) : 0u);
            const unsigned n1 = qdpp_u<Q_NEXT>(mine), n2 = qdpp_u<Q_DIAG>(mine), n3 = qdpp_u<Q_PREV>(mine);   // lane 3 reads lanes 0, 1, 2
            unsigned nb = bl ? (n1 | (n2 << 6) | (n3 << 12) | (mine << 18)) : 0u;
            const unsigned k1 = qdpp_u<Q_NEXT>(kickbits), k2 = qdpp_u<Q_DIAG>(kickbits), k3 = qdpp_u<Q_PREV>(kickbits);
            const unsigned kicks = k1 | (k2 << 6) | (k3 << 12) | (kickbits << 18);   // (lane 3's view: bit = robot index)
            if (__any(nb != 0)) {
                RSX_QS(7, 1);
#ifdef RSX_QSTATS
                { const unsigned mx = qs_wave_max((unsigned)__builtin_popcount(nb)); RSX_QS(8, mx); }
#endif
                float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f, bw = 0.0f;
                bool got = false;
                while (nb) {
                    const int k = __builtin_ctz(nb);
                    nb &= nb - 1;
                    const float4 oa = sh.A[k][q];
                    const float2 cs = sh.CS[k][q];
                    const float4 oc = make_float4(sh.W[k][q], cs.x, cs.y, ((kicks >> k) & 1u) ? 5.0f : 0.0f);   // om, c, s, kick speed
                    float dx = ball.x - oa.x, dy = ball.y - oa.y;
                    float nx = 0.0f, ny = 0.0f, pen = -1.0f;
                    bool mouth = false, touch = false;
                    float lx = fma_(dx, oc.y, dy * oc.z), ly = fma_(dy, oc.y, -(dx * oc.z));
                    if (fabsf(ly) < K::half_kw && lx > 0.0f) {
                        mouth = true; pen = K::dck_rb - lx; nx = oc.y; ny = oc.z; touch = pen > 0.0f;
                    } else {
                        float d2 = fma_(dx, dx, dy * dy);
                        if (d2 < K::rs_rb2 && d2 > 0.0f) {
                            float d = sqrtf(d2), inv = 1.0f / d;
                            nx = dx * inv; ny = dy * inv; pen = K::rs_rb - d; touch = true;
                        }
                    }
                    if (touch) {
                        got = true;
                        const float dvx = ball.vx - oa.z, dvy = ball.vy - oa.w;
                        float vn = fma_(dvx, nx, dvy * ny);
                        if (vn < 0.0f) {
                            const float wsum = fma_(ball.om, K::r_ball, oc.x * (mouth ? K::dck : K::r_robot));
                            const float vt = fma_(dvy, nx, -(dvx * ny)) - wsum;
                            float qb = K::ope_rb * vn * K::w_rb_b;
                            const float limb = qb * K::mu_rb;
                            const float ftb = clampf(vt * K::kt_rb_b, limb, -limb);
                            b0 = b0 - fma_(-ftb, ny, qb * nx); b1 = b1 - fma_(ftb, nx, qb * ny); bw = bw + ftb * K::spin_c;
                        }
                        float pb = K::beta * pen * K::w_rb_b;
                        b2 = b2 + pb * nx; b3 = b3 + pb * ny;
                    }
                    if (sweep == 0 && mouth && pen > -K::ir_tol && oc.w > 0.0f) {   // infrared + a kick command (no dribbler, no chip in this task)
                        bo.ovr = true; bo.okick = true;
                        bo.ovx = oa.z + oc.w * oc.y; bo.ovy = oa.w + oc.w * oc.z; bo.ovz = 0.0f;
                    }
                }
                if (got) {
                    ball.vx = ball.vx + b0; ball.vy = ball.vy + b1;
                    ball.x = ball.x + b2; ball.y = ball.y + b3;
                    ball.om = ball.om + bw;
                }
            }
            const unsigned dm = (deep ? 1u : 0u) | (wallp ? 2u : 0u);
            const unsigned de = (dm | qdpp_u<Q_NEXT>(dm)) | (qdpp_u<Q_DIAG>(dm) | qdpp_u<Q_PREV>(dm));
            deep_env = (de & 1u) != 0u; wall_env = (de & 2u) != 0u;
            wave_sync();   // every lane has read the snapshot before it is republished
        }
        if (bo.ovr) {   // kicker: decided in the first sweep, applied after the impulses
            ball.vx = bo.ovx; ball.vy = bo.ovy; ball.om = 0.0f;
            if (bo.okick && bo.ovz > 0.0f) ball.vz = bo.ovz;
        }
        // C: walls — only when some body of the wave is near one (near_walls, rsx_body.hpp: exact, the clamp is the identity elsewhere)
        {
            bool nw = near_walls_exact<KIND>(P, ball.x, ball.y);
#pragma unroll
            for (int m = 0; m < R; ++m) nw |= near_walls_exact<KIND>(P, r[m].x, r[m].y);
            if (__any(nw)) {
#pragma unroll
                for (int m = 0; m < R; ++m) { robot_walls<KIND>(P, r[m]); __builtin_amdgcn_sched_barrier(0); }
                ball_walls<KIND>(P, ball);
            }
        }
    }

    {   // what the rest of the step reads of the parameter block: fetched from the kernarg segment here instead of parked in VGPR lanes across the physics (like the SSL lane-group kernels, rsx_kernels.hpp)
        typedef const __attribute__((address_space(4))) uint32_t* kw_t;
        constexpr size_t KOFF = RSX_PARAMS_KERNARG_OFFSET;   // rsx_kernels.hpp, tied to RSX_HOT_ARGS by a static_assert
        kw_t pk = (kw_t)__builtin_amdgcn_kernarg_segment_ptr() + KOFF / 4;
        asm volatile("" : "+s"(pk));
        struct Words { uint32_t w[sizeof(Params) / 4]; } raww;
#pragma unroll
        for (size_t i = 0; i < sizeof(Params) / 4; ++i) raww.w[i] = pk[i];
        P = __builtin_bit_cast(Params, raww);
        RSX_UNPACK_HOT(P);
    }
    // ---- wire-format values, state rows, observation ----
    if (bl) {   // episode bookkeeping: fetched now (nothing of it is live during the physics)
        steps = __float_as_int(ld(A, ROW_STEPS * B4, eo));
        episode = __float_as_uint(ld(A, ROW_EPISODE * B4, eo));
        info[0] = ld(A, (ROW_INFO + 0) * B4, eo); info[1] = ld(A, (ROW_INFO + 1) * B4, eo);
        ep_ret = ld(A, ROW_EP_RET * B4, eo);
    }
    const __amdgpu_buffer_rsrc_t O = __builtin_amdgcn_make_buffer_rsrc(bufs.obs, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t F = __builtin_amdgcn_make_buffer_rsrc(bufs.final_obs, 0, -1, 0x00020000);
    const uint32_t oo = (uint32_t)Q_OD * eo + 8u + (uint32_t)(8 * R) * (uint32_t)p;   // byte offset of this lane's first robot in the env's observation row (eo = 4 e)
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    auto emit_obs = [&](const __amdgpu_buffer_rsrc_t rows) {   // positions only (README.md:88-90 style): [ball x y | robot x y ...]
        if (live) {
#pragma unroll
            for (int m = 0; m < R; ++m) {
                if (real(m)) {
                    const u2 v = {__builtin_bit_cast(unsigned, clampf(r[m].x * P.inv_max_pos, -1.2f, 1.2f)), __builtin_bit_cast(unsigned, clampf(r[m].y * P.inv_max_pos, -1.2f, 1.2f))};
                    __builtin_amdgcn_raw_buffer_store_b64(v, rows, (int)oo, 8 * m, 0);
                }
            }
            if (bl) {
                const u2 v = {__builtin_bit_cast(unsigned, clampf(ball.x * P.inv_max_pos, -1.2f, 1.2f)), __builtin_bit_cast(unsigned, clampf(ball.y * P.inv_max_pos, -1.2f, 1.2f))};
                __builtin_amdgcn_raw_buffer_store_b64(v, rows, (int)((uint32_t)Q_OD * eo), 0, 0);
            }
        }
    };
#pragma unroll
    for (int m = 0; m < R; ++m) {
        const float wd = r[m].om * K::rad2deg;
        float wh[4];
        wheel_speeds<KIND>(P, r[m], wh);   // from the carried (c, s) and the unrounded rate, like the other layouts
        r[m].om = wd * K::deg2rad;
        sincos_f32(r[m].th * K::deg2rad, r[m].s, r[m].c);
        if (live && real(m)) {
            const int p0 = (RS * m) * B4;
            stf(S, p0, ro, r[m].x); stf(S, p0 + B4, ro, r[m].y); stf(S, p0 + 2 * B4, ro, r[m].th); stf(S, p0 + 3 * B4, ro, r[m].vx);
            stf(S, p0 + 4 * B4, ro, r[m].vy); stf(S, p0 + 5 * B4, ro, wd);
            stf(S, p0 + 6 * B4, ro, ((irbits >> m) & 1u) ? 1.0f : 0.0f);
#pragma unroll
            for (int i = 0; i < 4; ++i) stf(S, p0 + (7 + i) * B4, ro, wh[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    ball.z = (K::r_ball + ball.z) - K::r_ball;
    emit_obs(O);

    // ---- reward, termination (README.md:96-102 style: a goal ends the episode) — the ball's lane ----
    float reward = 0.0f; int term = 0, trunc = 0;
    const bool first_step = steps == 0;
    if (first_step) { info[0] = 0.0f; info[1] = 0.0f; ep_ret = 0.0f; }
    bool success = false, against = false;
    if (bl) {
        const float bx = ball.x, by = ball.y;
        if (bx > P.half_len && fabsf(by) < P.ghw) { reward = 1.0f; term = 1; info[0] += 1.0f; }
        else if (bx < -P.half_len && fabsf(by) < P.ghw) { reward = -1.0f; term = 1; info[1] += 1.0f; }
        success = info[0] > 0.0f; against = info[1] > 0.0f;
        ep_ret = ep_ret + reward;
        steps += 1;
        trunc = steps >= P.max_steps;
        if (live) {
            stf(A, (ROW_INFO + 0) * B4, eo, info[0]); stf(A, (ROW_INFO + 1) * B4, eo, info[1]);
            stf(A, ROW_REWARD * B4, eo, reward);
            const __amdgpu_buffer_rsrc_t FL = __builtin_amdgcn_make_buffer_rsrc(bufs.flags, 0, -1, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)term, FL, (int)(eo >> 2), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)trunc, FL, (int)(eo >> 2), P.num_envs, 0);
        }
    }

    // ---- episode end: same-step auto-reset; every body places itself (one Philox block per body) ----
    const unsigned endm = (bl && (term | trunc)) ? 1u : 0u;
    const bool ended_env = qdpp_u<Q_L3>(endm) != 0u;
    bool new_episode = false;
    if (__any(ended_env)) {
        if (ended_env) emit_obs(F);   // terminal observation
        const uint32_t ep_new = qdpp_u<Q_L3>(episode) + 1u;   // the new episode's id, in all four lanes
        uint32_t eo_r = eo;
        asm volatile("" : "+v"(eo_r));   // an opaque copy: otherwise the first Philox round of the step's draws (same counter word) stays alive across the physics for this rare path
        const uint32_t env_id_r = P.env_id_base + (eo_r >> 2);
        if (ended_env) {
            if (bl) {
                episode = ep_new; new_episode = true;
                if (live) {
                    unsigned long long* const ms = metric_slot(bufs);
                    atomicAdd(&ms[1], 1ull);
                    if (success) atomicAdd(&ms[2], 1ull);
                    if (against) atomicAdd(&ms[3], 1ull);
                    atomicAdd(&ms[4], (unsigned long long)__float2ll_rn(ep_ret * 1048576.0f));
                    atomicAdd(&ms[5], (unsigned long long)steps);
                    if (trunc && !term) atomicAdd(&ms[6], 1ull);
                }
                steps = 0;
                const u32x4 u = philox4x32(env_id_r, ep_new, (uint32_t)N, DOM_PLACE, P.key0, P.key1);
                ball = Body{};
                ball.x = P.sc_jb * (u01(u.x) * 2.0f - 1.0f); ball.y = P.sc_jb * (u01(u.y) * 2.0f - 1.0f);
            }
#pragma unroll
            for (int m = 0; m < R; ++m) {   // robot k in cell (k % 6, k / 6) of a 6 x 4 grid: slot m of lane p is cell (m, p)
                const uint32_t b = (uint32_t)(R * p + m);
                const u32x4 u = philox4x32(env_id_r, ep_new, b, DOM_PLACE, P.key0, P.key1);
                const float jx = u01(u.x) * 2.0f - 1.0f, jy = u01(u.y) * 2.0f - 1.0f;
                const bool rl = real(m);
                r[m] = Body{};
                r[m].x = rl ? P.sc_sx * ((float)m - 2.5f) + P.sc_j * jx : qnan;
                r[m].y = rl ? P.sc_sy * ((float)p - 1.5f) + P.sc_j * jy : qnan;
                r[m].th = 360.0f * u01(u.z);
                sincos_f32(r[m].th * K::deg2rad, r[m].s, r[m].c);
                if (live && rl) {   // this env's rows were written before the reset was known
                    const int p0 = (RS * m) * B4;
                    stf(S, p0, ro, r[m].x); stf(S, p0 + B4, ro, r[m].y); stf(S, p0 + 2 * B4, ro, r[m].th); stf(S, p0 + 3 * B4, ro, 0.0f);
                    stf(S, p0 + 4 * B4, ro, 0.0f); stf(S, p0 + 5 * B4, ro, 0.0f); stf(S, p0 + 6 * B4, ro, 0.0f);
#pragma unroll
                    for (int i = 0; i < 4; ++i) stf(S, p0 + (7 + i) * B4, ro, 0.0f);
                }
            }
            emit_obs(O);
        }
    }

    // ---- store: the ball and the episode bookkeeping (the ball's lane) ----
    if (bl && live) {
        stf(S, 0, eo, ball.x); stf(S, B4, eo, ball.y); stf(S, 2 * B4, eo, K::r_ball + ball.z); stf(S, 3 * B4, eo, ball.vx); stf(S, 4 * B4, eo, ball.vy);
        stf(S, P.state_dim * B4, eo, ball.vz);
        stf(S, (P.state_dim + 1) * B4, eo, ball.om);
        stf(A, ROW_STEPS * B4, eo, __int_as_float(steps));
        if (new_episode) stf(A, ROW_EPISODE * B4, eo, __uint_as_float(episode));
        stf(A, ROW_EP_RET * B4, eo, ep_ret);
    }
    if (counts_steps) bufs.metrics[0] = bufs.metrics[0] + (unsigned long long)P.num_envs;
}

}  // namespace rsx
