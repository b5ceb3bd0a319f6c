// rsx_quad_ssl.hpp — the SSL 11v11 scrimmage task (BASELINE.json configs[3]: 22 robots + ball on the division-A
// field, every robot commanded on the device) for LARGE batches: "four lanes per env, six robots per lane".
//
// Same path, same buffers, same arithmetic as task_step_kernel<SSL, 32, SCRIMMAGE, 22, ...> (rsx_kernels.hpp; the
// reference's hot path: robosim.SSL.step / get_state, rsoccer_gym/Simulators/rsim.py:128-158, under a README.md:78-110
// style task) — other mapping: a DPP quad (lanes 4q .. 4q+3) owns env q of a 16-env tile; lane p holds the robots
// 6p .. 6p+5 in registers (lane 3: robots 18..21, its last two slots are ghosts at NaN positions that can touch
// nothing, and the ball).  What it buys over 32 lanes per env (one body per lane, 9 lanes idle, every pair tested
// from both sides through LDS): all lanes busy with robots, the per-step work (Philox block, targets, wheel speeds,
// eleven row stores per robot) issued once for sixteen envs instead of two, pair tests on registers — the other
// lanes' positions are DPP quad_perm operands of the subtraction, no LDS on the path without contacts.  What it costs:
// a lane walks six robots one after the other, so it needs enough envs to fill the chip (the host picks it from
// RSX_QUAD_MIN_ENVS on).  Results are bit-identical: per-body formulas are rsx_body.hpp's, every body sums its
// partners in index order (robots, then the ball), the ball sums the robots' records in robot order, draws use the same
// Philox counters (tests/test_gpu_parity.py::test_quad_layout_is_bit_identical).
#pragma once
#include "rsx_kernels.hpp"

#ifndef RSX_QUAD_WAVES
#define RSX_QUAD_WAVES 3   // waves per SIMD the kernel is compiled for
#endif
#ifndef RSX_QUAD_HOT_SLOTS
#define RSX_QUAD_HOT_SLOTS 5   // of 6 robot slots with a partner somewhere in the wave: skip the screening pass in the next sub-step (7 = never)
#endif

namespace rsx {

constexpr int Q_N = 22;        // robots
constexpr int Q_R = 6;         // robot slots per lane
constexpr int Q_ENVS = 16;     // envs per wave
constexpr int Q_OD = 2 + 2 * Q_N;   // observation width of the scrimmage task

struct QuadShared {
    // snapshot of a contact sweep, slot = body index (0..21 robots, 22 ball), column = env of the wave:
    // A = (x, y, vx, vy), C = (yaw rate | ball spin, cos, sin, kick speed); read by index on the contact path only
    float4 A[Q_N + 1][Q_ENVS];
    float4 C[Q_N + 1][Q_ENVS];
};

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
    Params P = P_; P.num_envs = hp_num_envs; P.state_dim = hp_state_dim;
    Buffers bufs = bufs_; bufs.state = hp_state; bufs.aux = hp_aux; bufs.actions = hp_in; bufs.flags = hp_flags;
    __shared__ QuadShared sh;
    const int lane = threadIdx.x;
    const int q = lane >> 2;           // env of the wave
    const int p = lane & 3;            // which quarter of the robots
    const bool bl = p == 3;            // the ball's lane (reward, termination, episode bookkeeping of the env)
    const int tile = tile_of_block_zigzag(hp_per_xcd);
    const int e_raw = tile * Q_ENVS + q;
    int live_i = e_raw < P.num_envs ? 1 : 0;
    asm volatile("" : "+v"(live_i));               // decided here: one flag through the step, not the index it is made of
    const bool live = live_i != 0;
    const int e = live ? e_raw : P.num_envs - 1;   // lanes beyond the batch shadow its last env (loads valid, stores masked)
    const uint32_t env_id = P.env_id_base + (uint32_t)e;
    // buffer addressing (rsx_epl.hpp): resource per array, row as the scalar offset, one 32-bit lane offset per array
    const uint32_t eo = 4u * (uint32_t)e;                                              // the env's column
    const uint32_t ro = eo + 4u * (uint32_t)(5 + RS * R * p) * (uint32_t)P.num_envs;     // ... from this lane's first robot row
    const __amdgpu_buffer_rsrc_t S = __builtin_amdgcn_make_buffer_rsrc(bufs.state, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t A = __builtin_amdgcn_make_buffer_rsrc(bufs.aux, 0, -1, 0x00020000);
    const int B4 = 4 * P.num_envs;
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
        const uint32_t rom = (m >= N - R * 3 && bl) ? ro - 4u * (uint32_t)(RS * 2) * (uint32_t)P.num_envs : ro;
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

    const uint32_t t = P.tick_base;
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
    bool hot = false;      // wave-uniform: the previous sub-step found (nearly) every robot slot in contact somewhere in the wave
    unsigned irbits = 0;   // infrared flags of this lane's robots (bit m), refreshed by the first sweep of every sub-step
    for (int sub = 0; sub < P.n_sub; ++sub) {
        // A: actuation + integration
#pragma unroll
        for (int m = 0; m < R; ++m) { integrate_robot<KIND>(P, r[m]); __builtin_amdgcn_sched_barrier(0); }
        integrate_ball<KIND>(P, ball);

        // B: contacts.  Fast path: the smallest squared distance of (my robots x my robots), (mine x the next lane's),
        // (mine x the lane after that) — together the four lanes cover every robot pair — and of (mine x ball).
        const bool ball_low = qdpp_f<Q_L3>(ball.z) < K::robot_h;
        constexpr uint32_t T_RR = __builtin_bit_cast(uint32_t, K::rs_rr2) - 1u;
        constexpr float NEAR2 = 0.13f * 0.13f;   // > (dck_rb + ir_tol)^2 + half_kw^2 = 0.126^2 and > rs_rb^2 (rsx_epl_ssl.hpp)
        irbits = 0;
        BallOverride bo{false, false, 0.0f, 0.0f, 0.0f};
        bool deep_env = false;
        for (int sweep = 0; sweep < 2; ++sweep) {
            const bool active = sweep == 0 || deep_env;
            auto dist2 = [](float xj, float yj, float xi, float yi) -> float {
                const float dx = xj - xi, dy = yj - yi;
                return fma_(dx, dx, dy * dy);
            };
            // the env's ball in all four lanes (DPP reads need the source lane ACTIVE: all cross-lane reads of a sweep
            // happen here and at its other wave-uniform points, never inside a lane-divergent branch)
            const float bx = qdpp_f<Q_L3>(ball.x), by = qdpp_f<Q_L3>(ball.y);
            const float bvx = qdpp_f<Q_L3>(ball.vx), bvy = qdpp_f<Q_L3>(ball.vy), bom = qdpp_f<Q_L3>(ball.om);
            // per robot of mine: does it touch anything?  rmin[i] = its smallest squared distance to my other robots and to the
            // robots of the next two lanes; cmin[j] = the smallest distance of the NEXT lane's robot j to mine (that lane does
            // not test backwards: it receives the flag); one robot of the other lanes at a time (two DPP temporaries)
            unsigned tf = 0, nf = 0;   // bit i: my robot i is closer than two radii to something / is near the ball
            if (!hot) {
            float rmin[R];
            unsigned cf = 0;   // bit j: the next lane's robot j is within two radii of one of mine
#pragma unroll
            for (int i = 0; i < R; ++i) rmin[i] = 1.0e30f;
#pragma unroll
            for (int j = R - 1; j >= 0; --j) {   // highest first: the flags are shifted in from the right
                const float xn = qdpp_f<Q_NEXT>(r[j].x), yn = qdpp_f<Q_NEXT>(r[j].y), xd = qdpp_f<Q_DIAG>(r[j].x), yd = qdpp_f<Q_DIAG>(r[j].y);
                float cn = 1.0e30f;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const float dn = dist2(xn, yn, r[i].x, r[i].y);
                    rmin[i] = fminf(rmin[i], fminf(dn, dist2(xd, yd, r[i].x, r[i].y)));
                    cn = fminf(cn, dn);
                    if (i < j) { const float di = dist2(r[j].x, r[j].y, r[i].x, r[i].y); rmin[i] = fminf(rmin[i], di); rmin[j] = fminf(rmin[j], di); }
                }
                cf = (cf + cf) + (unsigned)(cn < K::rs_rr2);
                nf = (nf + nf) + (unsigned)(ball_low & (dist2(bx, by, r[j].x, r[j].y) < NEAR2));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = R - 1; i >= 0; --i) tf = (tf + tf) + (unsigned)(rmin[i] < K::rs_rr2);
            tf |= qdpp_u<Q_PREV>(cf);          // what the previous lane found about my robots
            } else {
                // a wave whose previous sub-step had nearly every robot slot in contact (a scrum): the screening above would
                // flag them all again — every real robot goes to the exact partner test below instead (an empty set = no walk)
                tf = p == 3 ? 0xFu : 0x3Fu;
#pragma unroll
                for (int j = R - 1; j >= 0; --j) nf = (nf + nf) + (unsigned)(ball_low & (dist2(bx, by, r[j].x, r[j].y) < NEAR2));
            }
            if (!active) { tf = 0; nf = 0; }
            if (!__any((tf | nf) != 0)) { if (sweep == 0) hot = false; break; }

            // ---- some env of the wave has a contact: snapshot -> LDS; then robot slot by robot slot (only the slots that
            // are flagged in some env of the wave): partner set, walk, the robot's side of its ball contact ----
#pragma unroll
            for (int m = 0; m < R; ++m) {
                if (real(m)) {
                    sh.A[R * p + m][q] = make_float4(r[m].x, r[m].y, r[m].vx, r[m].vy);
                    sh.C[R * p + m][q] = make_float4(r[m].om, r[m].c, r[m].s, ((kickbits >> m) & 1u) ? 5.0f : 0.0f);
                }
            }
            if (bl) { sh.A[N][q] = make_float4(ball.x, ball.y, ball.vx, ball.vy); sh.C[N][q] = make_float4(ball.om, 0.0f, 0.0f, 0.0f); }
            wave_sync();
            auto key = [](float xj, float yj, float xi, float yi) -> uint32_t {   // exact integer form of 0 < d2 < thr (rsx_kernels.hpp)
                const float dx = xj - xi, dy = yj - yi;
                return __float_as_uint(fma_(dx, dx, dy * dy)) - 1u;
            };
            const unsigned near = nf;
            bool deep = false;
            unsigned rb_touch = 0;   // my robots that touch the ball in this sweep
            // partner sets first, from the snapshot positions (Jacobi: nothing has moved yet) — only for the robot slots
            // that are flagged in some env of the wave.  Bit = robot index: pushed from the highest relative position down
            // (previous lane's robots 5..0, the lane after the next, the next lane, my own), then rotated into place by 6 p
            unsigned todo[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                todo[i] = 0u;
                if (__any((tf >> i) & 1u)) {
                    unsigned rel = 0;
#define RSX_QPUSH(X, Y, SELF)                                                                                            \
                    _Pragma("unroll") for (int j = R - 1; j >= 0; --j) {                                                   \
                        const uint32_t u = ((SELF) && j == i) ? 0xFFFFFFFFu : key((X), (Y), r[i].x, r[i].y);             \
                        asm("v_cmp_gt_u32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(rel) : "v"(u), "s"(T_RR) : "vcc"); \
                    }
                    RSX_QPUSH(qdpp_f<Q_PREV>(r[j].x), qdpp_f<Q_PREV>(r[j].y), false)
                    RSX_QPUSH(qdpp_f<Q_DIAG>(r[j].x), qdpp_f<Q_DIAG>(r[j].y), false)
                    RSX_QPUSH(qdpp_f<Q_NEXT>(r[j].x), qdpp_f<Q_NEXT>(r[j].y), false)
                    RSX_QPUSH(r[j].x, r[j].y, true)
#undef RSX_QPUSH
                    const unsigned sh6 = 6u * (unsigned)p;
                    todo[i] = ((tf >> i) & 1u) ? (((rel << sh6) | (rel >> (24u - sh6))) & 0xFFFFFFu) : 0u;
                }
            }
            if (sweep == 0) {   // how many robot slots really have a partner somewhere in the wave decides the next sub-step's mode
                int slots = 0;
#pragma unroll
                for (int i = 0; i < R; ++i) slots += __any(todo[i] != 0u) ? 1 : 0;
                hot = slots >= RSX_QUAD_HOT_SLOTS;
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {   // each robot of the lane: its robot partners in index order, then the ball
                if (__any(((tf | near) >> i) & 1u)) {
                    const unsigned todo_i = todo[i];
                    float avx = 0.0f, avy = 0.0f, apx = 0.0f, apy = 0.0f, unused = 0.0f;
                    unsigned td = todo_i;
                    while (td) {
                        const int j = __builtin_ctz(td);
                        td &= td - 1;
                        const float4 oj = sh.A[j][q];
                        const float wj = sh.C[j][q].x;
                        const float dx = oj.x - r[i].x, dy = oj.y - r[i].y;
                        contact_response(r[i], oj, fma_(dx, dx, dy * dy), K::rs_rr, K::ope_rr, K::w_rr, K::kt_rr, K::mu_rr, 0.0f,
                                         fma_(wj, K::r_robot, r[i].om * K::r_robot), K::beta, K::pen2, avx, avy, apx, apy, unused, deep);
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
                    if (todo_i | (touch ? 1u : 0u)) {   // only a body that touched something is updated
                        r[i].vx = r[i].vx + avx; r[i].vy = r[i].vy + avy;
                        r[i].x = r[i].x + apx; r[i].y = r[i].y + apy;
                    }
                }
            }
            // the ball's side: the robots near it, in robot order, from the snapshot — the same expressions the robot's lane
            // evaluated (its own view of the contact), so the same numbers; kicker: the last robot in index order wins
            const unsigned n1 = qdpp_u<Q_NEXT>(near), n2 = qdpp_u<Q_DIAG>(near), n3 = qdpp_u<Q_PREV>(near);   // lane 3 reads lanes 0, 1, 2
            unsigned nb = bl ? (n1 | (n2 << 6) | (n3 << 12) | (near << 18)) : 0u;
            if (__any(nb != 0)) {
                float b0 = 0.0f, b1 = 0.0f, b2 = 0.0f, b3 = 0.0f, bw = 0.0f;
                bool got = false;
                while (nb) {
                    const int k = __builtin_ctz(nb);
                    nb &= nb - 1;
                    const float4 oa = sh.A[k][q];
                    const float4 oc = sh.C[k][q];   // om, c, s, kick
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
            (void)rb_touch;
            const unsigned dm = deep ? 1u : 0u;
            deep_env = ((dm | qdpp_u<Q_NEXT>(dm)) | (qdpp_u<Q_DIAG>(dm) | qdpp_u<Q_PREV>(dm))) != 0u;
            wave_sync();   // every lane has read the snapshot before it is republished
        }
        if (bo.ovr) {   // kicker: decided in the first sweep, applied after the impulses
            ball.vx = bo.ovx; ball.vy = bo.ovy; ball.om = 0.0f;
            if (bo.okick && bo.ovz > 0.0f) ball.vz = bo.ovz;
        }
        // C: walls — only when some body of the wave is near one (near_walls, rsx_body.hpp: exact, the clamp is the identity elsewhere)
        {
            bool nw = near_walls<KIND>(P, ball.x, ball.y);
#pragma unroll
            for (int m = 0; m < R; ++m) nw |= near_walls<KIND>(P, r[m].x, r[m].y);
            if (__any(nw)) {
#pragma unroll
                for (int m = 0; m < R; ++m) { robot_walls<KIND>(P, r[m]); __builtin_amdgcn_sched_barrier(0); }
                ball_walls<KIND>(P, ball);
            }
        }
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
                    __builtin_amdgcn_raw_buffer_store_b64(v, rows, (int)oo, 8 * m, RSX_OBS_AUX);
                }
            }
            if (bl) {
                const u2 v = {__builtin_bit_cast(unsigned, clampf(ball.x * P.inv_max_pos, -1.2f, 1.2f)), __builtin_bit_cast(unsigned, clampf(ball.y * P.inv_max_pos, -1.2f, 1.2f))};
                __builtin_amdgcn_raw_buffer_store_b64(v, rows, (int)((uint32_t)Q_OD * eo), 0, RSX_OBS_AUX);
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
