// rsx_epl_ssl.hpp — the four registered SSL tasks (SSLStaticDefenders-v0 1v6, SSLDribbling-v0 1v4,
// SSLContestedPossession-v0 1v1, SSLPassEndurance-v0 2v0) as fused steps in the "one lane per env" layout
// for LARGE batches: the SSL counterpart of rsx_epl.hpp.
//
// Same path, same buffers, same arithmetic as task_step_kernel<SSL, 8, TASK, N, ...>
// (reference: ssl/ssl_hw_challenge/{static_defenders,dribbling,contested_possession,pass_endurance}.py
// around robosim.SSL.step, rsim.py:155,158); the task arithmetic itself (agent commands, observation,
// reward / termination, placement) is the SAME code: ssl_agent_commands, write_obs_nb, task_reward and
// place_env of rsx_kernels.hpp.  Other mapping: lane = env, a wave owns 64 envs and walks the bodies of each
// one after the other.  The 8-lane layout runs its ball lane and its robot lanes one after the other and tests
// every robot pair from both sides; at scale that kernel is VALU-bound.  Results are bit-identical:
// every body sums its partners in index order (robot-robot pairs first, then the ball), the ball sums the robots'
// records in robot order, each side of a pair evaluates its own response with the same expressions, draws use
// the same Philox counters (tests/test_gpu_parity.py::test_env_per_lane_layout_is_bit_identical).
//
// What differs from the VSS kernel: holonomic actuation (only blue 0 is driven by the agent: the other robots
// hold still unless they are hit; pass endurance keeps the receiver's dribbler on), the kicker mouth / infrared /
// kick / dribbler of the robot-ball contact (evaluated only for robots whose centre is within 13 cm of the ball:
// a mouth or infrared contact needs < 12.6 cm), the ball's flight, SSL walls, eleven state rows per robot
// (infrared and four wheel speeds are outputs: written every step, read only when the step has no physics).
#pragma once
#include "rsx_epl_common.hpp"

#ifndef RSX_SD_STEP_WAVES
#define RSX_SD_STEP_WAVES 4   // waves per SIMD the LEAN 1v6 single-step kernel is compiled for (128 VGPRs, no scratch: 262 144 envs 51 -> 47 us)
#endif

namespace rsx {

template <int N>
struct SeplShared {
    // contact sums (column = lane); the same bytes are the pose scratch A[body][lane] of a reset placement
    // (place_env).  Observations stay in registers and go out as vector stores (see rsx_epl.hpp).
    union {
        EplSums<N + 1> c;
        float4 A[(N + 1) * 64];
    };
};

template <int TASK> struct SeplTask;   // robots, blue robots, observation width, robots that carry kicker / dribbler commands
template <> struct SeplTask<RSX_TASK_SSL_STATIC_DEFENDERS> { static constexpr int N = 7, NBLUE = 1, OD = 24, NCMD = 1, WMIN = 3, WMAX = 4; };   // static_defenders.py:47-48
template <> struct SeplTask<RSX_TASK_SSL_DRIBBLING> { static constexpr int N = 5, NBLUE = 1, OD = 21, NCMD = 1, WMIN = 4, WMAX = 8; };          // dribbling.py:45-46
template <> struct SeplTask<RSX_TASK_SSL_CONTESTED> { static constexpr int N = 2, NBLUE = 1, OD = 14, NCMD = 1, WMIN = 4, WMAX = 8; };          // contested_possession.py:46-47
template <> struct SeplTask<RSX_TASK_SSL_PASS_ENDURANCE> { static constexpr int N = 2, NBLUE = 2, OD = 16, NCMD = 2, WMIN = 4, WMAX = 8; };     // pass_endurance.py:45-46

// occupancy target (waves per SIMD): the 1v6 kernel holds 7 robots in registers (3-4 measured best); the smaller
// tasks fit 128 VGPRs (4 waves) without spilling — 6 and 8 waves spill and measured 1.4x / 2.3x slower, 3 the same
template <int TASK, int MODE, bool LEAN = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((TASK == RSX_TASK_SSL_STATIC_DEFENDERS && LEAN) ? RSX_SD_STEP_WAVES : SeplTask<TASK>::WMIN, SeplTask<TASK>::WMAX)))
void ssl_epl_kernel(RSX_HOT_ARGS, const Params P_, const Buffers bufs_) {
    constexpr int KIND = RSX_KIND_SSL, N = SeplTask<TASK>::N, NBLUE = SeplTask<TASK>::NBLUE, OD = SeplTask<TASK>::OD,
                  NCMD = SeplTask<TASK>::NCMD, RS = 11, NB1 = N + 1;
    constexpr bool HAS_TS = TASK == RSX_TASK_SSL_DRIBBLING || TASK == RSX_TASK_SSL_PASS_ENDURANCE;   // checkpoints_count / stopped_steps
    using K = KC<KIND>;
    using T = TC<TASK>;
    constexpr int ID = T::info_dim, AD = T::act_dim;
    constexpr bool STEP = MODE == MODE_STEP;
    static_assert(STEP || !LEAN, "the lean form is a single-step form");
    // LEAN (chosen by the launcher per task and batch): episode bookkeeping and the pre-step positions are fetched AFTER the
    // physics, robot rows are stored robot by robot while the observation is assembled — 151 -> 128 VGPRs for 1v6 (4 waves per
    // SIMD), at the price of a second set of row stores in the lanes that reset and of loads nothing can hide behind
    constexpr bool LATE = LEAN;
    Params P = P_; RSX_UNPACK_HOT(P);
    Buffers bufs = bufs_; bufs.state = hp_state; bufs.aux = hp_aux; bufs.actions = hp_in; bufs.flags = hp_flags;
    const int n_steps = MODE == MODE_ROLLOUT ? (hp_n_steps & RSX_N_STEPS_MASK) : 1;
    __shared__ SeplShared<N> sh;
    const int lane = threadIdx.x;
    const bool tick_dev = (hp_n_steps & RSX_TICK_DEV) != 0;   // step counter of this launch: rsx_kernels.hpp, step_tick
    const StepTick tk = step_tick(tick_dev, P, bufs, (uint32_t)n_steps);
    if (__builtin_expect(!tk.ok, 0)) return;
    const int tile = tile_of_block_zigzag(zigzag_per(tick_dev, tk.t, hp_per_xcd));
    const int e_raw = tile * 64 + lane;
    int live_i = e_raw < P.num_envs ? 1 : 0;
    asm volatile("" : "+v"(live_i));   // decided here: one flag through the step, not the index it is made of
    const bool live = live_i != 0;
    const int e = live ? e_raw : P.num_envs - 1;   // lanes beyond the batch shadow its last env (loads valid and unconditional; stores, counters masked)
    const size_t B = (size_t)P.num_envs;
    const uint32_t env_id = P.env_id_base + (uint32_t)e;
    EplIO io(bufs.state, bufs.aux, P.row_stride, e);   // this lane's column of the [rows][B] arrays (rsx_epl_common.hpp)
    asm volatile("" : "+v"(io.eo));   // THE per-lane offset of the step: everything later derives from it, not from the env index
    const uint32_t eo = io.eo;
    const __amdgpu_buffer_rsrc_t S = io.S, A = io.A;

    RSX_STAMP(0);
    // ---- load ----
    Body r[N];
    Body ball = Body{};
    float wdeg[N], wheels[N][4];
    float info[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float ep_ret = 0.0f, prev_pot = 0.0f;
    int steps = 0; uint32_t episode = 0;
    float raw[N][6], rawb[7] = {0, 0, 0, 0, 0, 0, 0};
    int ir_in[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int f = 0; f < 6; ++f) raw[k][f] = io.ld(S, 5 + RS * k + f);
        ir_in[k] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) wheels[k][i] = 0.0f;
        // infrared is refreshed by every step that has physics; a step without (time_step 0) keeps the stored flag
        if (P.n_sub == 0) ir_in[k] = io.ld(S, 5 + RS * k + 6) != 0.0f;
    }
    {
#pragma unroll
        for (int f = 0; f < 5; ++f) rawb[f] = io.ld(S, f);
        rawb[5] = io.ld(S, P.state_dim);
        rawb[6] = io.ld(S, P.state_dim + 1);
        if (!LATE) {   // the lean form fetches the episode bookkeeping after the physics (nothing before needs it)
            steps = __float_as_int(io.ld(A, ROW_STEPS));
            episode = __float_as_uint(io.ld(A, ROW_EPISODE));
#pragma unroll
            for (int i = 0; i < ID; ++i) info[i] = io.ld(A, ROW_INFO + i);
            ep_ret = io.ld(A, ROW_EP_RET);
            if (HAS_TS) prev_pot = io.ld(A, ROW_PREV_POT);
        }
    }
    const bool counts_steps = blockIdx.x == 0 && lane == 0;   // metrics[0]: see task_step_kernel (single-step: read-modify-write at the end, lane re-derived there)
    unsigned long long steps_before = 0;
    if (!LATE && counts_steps) steps_before = bufs.metrics[0];
    const bool fed = MODE == MODE_STEP && bufs.actions != nullptr;
    float act[5] = {0, 0, 0, 0, 0};
    if (fed) {
#pragma unroll
        for (int i = 0; i < AD; ++i) act[i] = bufs.actions[(size_t)e * AD + i];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): all loads land once, before the step loop
    RSX_STAMP(1);
#pragma unroll
    for (int k = 0; k < N; ++k) {   // interpret_body, robot
        r[k] = Body{};
        r[k].x = raw[k][0]; r[k].y = raw[k][1]; r[k].vx = raw[k][3]; r[k].vy = raw[k][4];
        r[k].th = raw[k][2];
        wdeg[k] = raw[k][5];
        r[k].om = raw[k][5] * K::deg2rad;
        r[k].ir = ir_in[k];
        sincos_f32(r[k].th * K::deg2rad, r[k].s, r[k].c);
    }
    ball.x = rawb[0]; ball.y = rawb[1]; ball.vx = rawb[3]; ball.vy = rawb[4];
    ball.z = rawb[2] - K::r_ball; ball.vz = rawb[5]; ball.om = rawb[6];

    float reward = 0.0f; int term = 0, trunc = 0;
    RSX_STAMP(2);   // development builds (-DRSX_TIMING, tools/exp_timeline_epl.py): cycle stamps of wave 0's lane 0 per block

    for (int it = 0; it < n_steps; ++it) {
        bool first_step = steps == 0;
        const uint32_t t = tk.t + (uint32_t)it;   // see task_step_kernel
        if (!LATE && first_step) {
#pragma unroll
            for (int i = 0; i < 10; ++i) info[i] = 0.0f;
            ep_ret = 0.0f;
        }
        float last_bx = ball.x, last_by = ball.y;        // the reference's last_frame (pre-step); single-step launches
        float last_rx = r[0].x, last_ry = r[0].y;        // read these four from the (not yet overwritten) rows after the physics
        float obs_ts = prev_pot;   // the task scalar as this step's observation sees it

        // ---- action -> commands: only blue 0 is driven by the agent ----
        {
            float a[5] = {0, 0, 0, 0, 0};
            const StepDraw dr = draw_for_step<KIND, TASK>(P, env_id, t, 0, true, fed);
#pragma unroll
            for (int i = 0; i < AD; ++i) a[i] = fed ? act[i] : dr.v[i];
            float q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            ssl_agent_commands<TASK>(a, r[0].s, r[0].c, q);
            robot_targets<KIND>(P, r[0], q);
#pragma unroll
            for (int k = 1; k < N; ++k) {
                float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (TASK == RSX_TASK_SSL_PASS_ENDURANCE && k == 1) z[7] = 1.0f;   // receiver: dribbler on
                robot_targets<KIND>(P, r[k], z);
            }
        }

        RSX_STAMP(3);
        // ---- physics: n_sub sub-steps, the whole env in registers ----
        ball_step_friction(P, ball);   // rolling resistance + spin decay, once per step() (rsx_body.hpp, like every per-body formula below)
        for (int sub = 0; sub < P.n_sub; ++sub) {
            // A: actuation + integration (holonomic)
#pragma unroll
            for (int k = 0; k < N; ++k) integrate_robot<KIND>(P, r[k]);
            integrate_ball<KIND>(P, ball);

            if (sub == 0) RSX_STAMP(4);
            // B: contacts.  One bit per touching robot pair (exact integer form of 0 < d2 < thr, see
            // rsx_kernels.hpp), one bit per robot whose centre is near enough to the ball for a mouth,
            // circle or infrared contact; a second sweep over the corrected snapshot where a pair was deep.
            const bool ball_low = ball.z < K::robot_h;
            constexpr uint32_t T_RR = __builtin_bit_cast(uint32_t, K::rs_rr2) - 1u;
            constexpr float NEAR2 = 0.13f * 0.13f;   // > (dck_rb + ir_tol)^2 + half_kw^2 = 0.126^2 and > rs_rb^2
            auto find_pairs = [&]() -> unsigned {
                // bit p = pair p touches: pairs visited in reverse order, each result shifted in from the right (a compare
                // and one add-with-carry per pair, no bit constant in a register; see rsx_epl.hpp)
                unsigned touching = 0;
#pragma unroll
                for (int i = N - 2; i >= 0; --i) {
#pragma unroll
                    for (int j = N - 1; j > i; --j) {
                        const float dx = r[j].x - r[i].x, dy = r[j].y - r[i].y;
                        const uint32_t u = __float_as_uint(fma_(dx, dx, dy * dy)) - 1u;
                        asm("v_cmp_gt_u32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(touching) : "v"(u), "s"(T_RR) : "vcc");
                    }
                }
                return touching;
            };
            auto find_near = [&]() -> unsigned {
                unsigned near = 0;
#pragma unroll
                for (int k = N - 1; k >= 0; --k) {
                    const float dx = ball.x - r[k].x, dy = ball.y - r[k].y;
                    const float d2 = fma_(dx, dx, dy * dy);
                    asm("v_cmp_gt_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(near) : "v"(d2), "s"(NEAR2) : "vcc");
                }
                return ball_low ? near : 0u;
            };
            bool deep = false, wallp = false;
            BallOverride bo{false, false, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < N; ++k) r[k].ir = 0;   // refreshed by the first sweep; robots far from the ball: no infrared
            for (int sweep = 0; sweep < 4; ++sweep) {   // the later sweeps run the same (cached) instructions
                // second: envs with a deep pair only; third and fourth: envs whose last sweep also saw a wall pair (model v2)
                const bool mine = sweep == 0 || (deep && (sweep == 1 || wallp));
                if (sweep >= 1 && !__any(mine)) break;   // no env of the wave goes on: no further pair test either
                const unsigned touching = mine ? find_pairs() : 0u;
                const unsigned near = mine ? find_near() : 0u;
                if (sub == 0 && sweep == 0) RSX_STAMP(5);
                if (!__any((touching | near) != 0)) break;
                const bool first = sweep == 0;
                const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // = lane, re-derived: the index is not held across the sub-steps for this path
                epl_zero_sums(sh.c, ln);
                wave_sync();
                deep = false; wallp = false;
                // robot-robot pairs, in pair order: every body receives its partners in index order (rsx_epl_common.hpp)
                epl_walk_pairs<KIND, N, false>(P, r, ball, sh.c, ln, touching, deep, wallp);
                // robot-ball, robot by robot (the ball sums the robots' records in robot order): kicker mouth
                // (flat face at dck) or body circle; n points robot -> ball.  Mirrors ssl_sweep.
                unsigned rb_touch = 0;   // robots that touch the ball in this sweep
                unsigned nr = near;
                while (nr) {
                    const int k = __builtin_ctz(nr);
                    nr &= nr - 1;
                    Body o = Body{};
                    float kick_x = 0.0f, kick_z = 0.0f; int drib = 0;
#pragma unroll
                    for (int q = 0; q < N; ++q)
                        { const bool m = k == q; o.x = m ? r[q].x : o.x; o.y = m ? r[q].y : o.y; o.vx = m ? r[q].vx : o.vx; o.vy = m ? r[q].vy : o.vy; o.om = m ? r[q].om : o.om; o.c = m ? r[q].c : o.c; o.s = m ? r[q].s : o.s; }
#pragma unroll
                    for (int q = 0; q < NCMD; ++q)   // the other robots get zero commands
                        if (k == q) { kick_x = r[q].kick_x; kick_z = r[q].kick_z; drib = r[q].drib; }
                    float dx = ball.x - o.x, dy = ball.y - o.y;
                    float nx = 0.0f, ny = 0.0f, pen = -1.0f;
                    bool mouth = false, touch = false;
                    {
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
                    }
                    if (touch) {
                        rb_touch |= 1u << k;
                        deep |= pen > K::pen2;
                        float a0 = sh.c.acc[0][k][ln], a1 = sh.c.acc[1][k][ln], a2 = sh.c.acc[2][k][ln], a3 = sh.c.acc[3][k][ln];
                        float b0 = sh.c.acc[0][N][ln], b1 = sh.c.acc[1][N][ln], b2 = sh.c.acc[2][N][ln], b3 = sh.c.acc[3][N][ln];
                        float bw = sh.c.accw[ln];
                        const float dvx = ball.vx - o.vx, dvy = ball.vy - o.vy;
                        float vn = fma_(dvx, nx, dvy * ny);
                        if (vn < 0.0f) {
                            float q = K::ope_rb * vn * K::w_rb_r; a0 = fma_(q, nx, a0); a1 = fma_(q, ny, a1);
                            const float wsum = fma_(ball.om, K::r_ball, o.om * (mouth ? K::dck : K::r_robot));
                            const float vt = fma_(dvy, nx, -(dvx * ny)) - wsum;
                            const float lim = q * K::mu_rb;
                            const float ft = clampf(vt * K::kt_rb_r, lim, -lim);
                            a0 = fma_(-ft, ny, a0); a1 = fma_(ft, nx, a1);
                            // the ball's side of the same contact
                            float qb = K::ope_rb * vn * K::w_rb_b;
                            const float limb = qb * K::mu_rb;
                            const float ftb = clampf(vt * K::kt_rb_b, limb, -limb);
                            b0 = b0 - fma_(-ftb, ny, qb * nx); b1 = b1 - fma_(ftb, nx, qb * ny); bw = bw + ftb * K::spin_c;
                        }
                        float pc = K::beta * pen * K::w_rb_r;
                        a2 = fma_(-pc, nx, a2); a3 = fma_(-pc, ny, a3);
                        float pb = K::beta * pen * K::w_rb_b;
                        b2 = b2 + pb * nx; b3 = b3 + pb * ny;
                        sh.c.acc[0][k][ln] = a0; sh.c.acc[1][k][ln] = a1; sh.c.acc[2][k][ln] = a2; sh.c.acc[3][k][ln] = a3;
                        sh.c.acc[0][N][ln] = b0; sh.c.acc[1][N][ln] = b1; sh.c.acc[2][N][ln] = b2; sh.c.acc[3][N][ln] = b3;
                        sh.c.accw[ln] = bw;
                    }
                    if (first) {
                        const bool ir = mouth && pen > -K::ir_tol;
#pragma unroll
                        for (int q = 0; q < N; ++q) if (k == q) r[q].ir = ir;
                        if (ir) {  // infrared: kicker / dribbler act on the ball (the last robot in index order wins)
                            if (kick_x > 0.0f || kick_z > 0.0f) {
                                bo.ovr = true; bo.okick = true;
                                bo.ovx = o.vx + kick_x * o.c; bo.ovy = o.vy + kick_x * o.s; bo.ovz = kick_z;
                            } else if (drib) {
                                float hx = o.x + K::dck_rb * o.c, hy = o.y + K::dck_rb * o.s;
                                float cvx = (hx - ball.x) * P.drib_gain, cvy = (hy - ball.y) * P.drib_gain;
                                float m2 = cvx * cvx + cvy * cvy;
                                if (m2 > K::drib_vmax2) { float sc = K::drib_vmax / sqrtf(m2); cvx = cvx * sc; cvy = cvy * sc; }
                                bo.ovr = true; bo.okick = false;
                                bo.ovx = (o.vx - o.om * K::dck_rb * o.s) + cvx;
                                bo.ovy = (o.vy + o.om * K::dck_rb * o.c) + cvy;
                                bo.ovz = 0.0f;
                            }
                        }
                    }
                }
                wave_sync();
                epl_apply_sums<N>(r, ball, sh.c, ln, [&](int k) {   // only a body that touched something is updated
                    return k == N ? rb_touch != 0 : ((touching & epl_pair_mask<N>(k)) | (rb_touch & (1u << k))) != 0;
                });
                wave_sync();
            }
            if (sub == 0) RSX_STAMP(6);
            if (bo.ovr) {   // kicker / dribbler: decided in the first sweep, applied after the impulses
                ball.vx = bo.ovx; ball.vy = bo.ovy; ball.om = 0.0f;
                if (bo.okick && bo.ovz > 0.0f) ball.vz = bo.ovz;
            }
            // C: walls
            {   // only when some body of the wave is near a wall (near_walls, rsx_body.hpp: the clamp is the identity elsewhere)
                // (the exact gate where eight clamps are at stake — 1v6, 1v4; the three-instruction one for the two-robot tasks)
                constexpr bool XG = N >= 5;
                bool nw = XG ? near_walls_exact<KIND>(P, ball.x, ball.y) : near_walls<KIND>(P, ball.x, ball.y);
#pragma unroll
                for (int k = 0; k < N; ++k) nw |= XG ? near_walls_exact<KIND>(P, r[k].x, r[k].y) : near_walls<KIND>(P, r[k].x, r[k].y);
                if (__any(nw)) {
#pragma unroll
                    for (int k = 0; k < N; ++k) robot_walls<KIND>(P, r[k]);
                    ball_walls<KIND>(P, ball);
                }
            }
            if (sub == 0) RSX_STAMP(7);
        }
        RSX_STAMP(8);

        // ---- wire-format values, observation, reward ----
        if (LATE) {   // episode bookkeeping, cumulative terms and the pre-step positions: fetched now (the rows still hold the pre-step state)
            steps = __float_as_int(io.ld(A, ROW_STEPS));
            episode = __float_as_uint(io.ld(A, ROW_EPISODE));
#pragma unroll
            for (int i = 0; i < ID; ++i) info[i] = io.ld(A, ROW_INFO + i);
            ep_ret = io.ld(A, ROW_EP_RET);
            if (HAS_TS) { prev_pot = io.ld(A, ROW_PREV_POT); obs_ts = prev_pot; }
            last_bx = io.ld(S, 0); last_by = io.ld(S, 1);
            last_rx = io.ld(S, 5); last_ry = io.ld(S, 6);
            first_step = steps == 0;
            if (first_step) {
#pragma unroll
                for (int i = 0; i < 10; ++i) info[i] = 0.0f;
                ep_ret = 0.0f;
            }
        }
        float ob[OD];   // this env's observation, in registers
        float w0[4] = {0, 0, 0, 0};   // robot 0's wheel speeds (reward terms)
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float wd = r[k].om * K::rad2deg;
            wdeg[k] = wd;
            wheel_speeds<KIND>(P, r[k], wheels[k]);   // from the carried (c, s) and the unrounded rate, like the other layout
            if (k == 0) { w0[0] = wheels[0][0]; w0[1] = wheels[0][1]; w0[2] = wheels[0][2]; w0[3] = wheels[0][3]; }
            r[k].om = wd * K::deg2rad;
            sincos_f32(r[k].th * K::deg2rad, r[k].s, r[k].c);
            write_obs_nb<KIND, TASK>(P, ob, k, NBLUE, true, false, r[k].x, r[k].y, r[k].vx, r[k].vy, r[k].s, r[k].c, wd, r[k].ir, obs_ts);
            if (LATE && live) {   // wire format, robot by robot (an env that resets below writes its rows again)
                io.st_robot(5 + RS * k, r[k].x, r[k].y, r[k].th, r[k].vx, r[k].vy, wd);
                io.st(S, 5 + RS * k + 6, r[k].ir ? 1.0f : 0.0f);
#pragma unroll
                for (int i = 0; i < 4; ++i) io.st(S, 5 + RS * k + 7 + i, wheels[k][i]);
            }
            if (LATE) __builtin_amdgcn_sched_barrier(0);
        }
        RSX_STAMP(9);
        ball.z = (K::r_ball + ball.z) - K::r_ball;
        write_obs_nb<KIND, TASK>(P, ob, N, NBLUE, false, true, ball.x, ball.y, ball.vx, ball.vy, 0.0f, 0.0f, 0.0f, 0, obs_ts);
        bool success = false, against = false;
        {   // what the reward needs from the robots, in the slots task_step_kernel uses
            float xr[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            xr[0] = r[0].x; xr[1] = r[0].y;
            if (TASK == RSX_TASK_SSL_STATIC_DEFENDERS || TASK == RSX_TASK_SSL_CONTESTED) {
                xr[6] = last_rx; xr[7] = last_ry;
                xr[8] = w0[0]; xr[9] = w0[1]; xr[10] = w0[2]; xr[11] = w0[3];
            }
#pragma unroll
            for (int k = 1; k < N; ++k) {
                if (TASK == RSX_TASK_SSL_DRIBBLING) xr[1 + k] = (fabsf(r[k].vx) > 0.05f || fabsf(r[k].vy) > 0.05f) ? 1.0f : 0.0f;
                if (TASK == RSX_TASK_SSL_CONTESTED && k == 1) xr[2] = (fabsf(r[k].vx) > 0.1f || fabsf(r[k].vy) > 0.1f) ? 1.0f : 0.0f;
                if (TASK == RSX_TASK_SSL_PASS_ENDURANCE && k == 1) { xr[2] = r[k].x; xr[3] = r[k].y; xr[4] = r[k].ir ? 1.0f : 0.0f; }
            }
            task_reward<KIND, TASK>(P, xr, ball.x, ball.y, last_bx, last_by, first_step, prev_pot, info, reward, term, success, against);
            ep_ret = ep_ret + reward;
        }
        steps += 1;
        trunc = steps >= P.max_steps;
        const bool ended = live && (term | trunc);
        if (live) {
#pragma unroll
            for (int i = 0; i < ID; ++i) io.st(A, ROW_INFO + i, info[i]);
            io.st(A, ROW_REWARD, reward);
            io.st_flags(bufs.flags, P.num_envs, term, trunc);
        }

        RSX_STAMP(10);
        // ---- episode end: same-step auto-reset, one lane = one env ----
        if (__any(ended)) {
            if (ended) {
                epl_store_row<OD>(bufs.final_obs, eo, ob);   // terminal observation
                episode += 1;
                unsigned long long* const ms = metric_slot(bufs);
                atomicAdd(&ms[1], 1ull);
                if (success) atomicAdd(&ms[2], 1ull);
                atomicAdd(&ms[4], (unsigned long long)__float2ll_rn(ep_ret * 1048576.0f));
                atomicAdd(&ms[5], (unsigned long long)steps);
                if (trunc && !term) atomicAdd(&ms[6], 1ull);
                // placement: the task's reset (place_env: sequential, Philox draws), poses through this lane's
                // column of the (now idle) contact sums
                // (env id and lane re-derived here: an opaque copy keeps the first Philox round of the step's draws — same counter
                // word — and the lane index from staying alive across the physics for this rare path)
                uint32_t eo_r = eo;
                asm volatile("" : "+v"(eo_r));
                const uint32_t env_id_r = P.env_id_base + (eo_r >> 2);
                const int lane_r = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // = lane (64-thread workgroups)
                place_env<TASK, 1, false>(P, N, env_id_r, episode, lane_r, sh.A, nullptr);
                steps = 0;
                if (HAS_TS) prev_pot = 0.0f;
                float4 np[NB1];
#pragma unroll
                for (int k = 0; k < NB1; ++k) np[k] = sh.A[k * 64 + lane_r];
                wave_sync();   // the scratch is the contact sums again from here on
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    r[k] = Body{};
                    r[k].x = np[k].x; r[k].y = np[k].y;
                    r[k].th = np[k].z;
                    wdeg[k] = 0.0f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) wheels[k][i] = 0.0f;
                    sincos_f32(r[k].th * K::deg2rad, r[k].s, r[k].c);
                    write_obs_nb<KIND, TASK>(P, ob, k, NBLUE, true, false, r[k].x, r[k].y, r[k].vx, r[k].vy, r[k].s, r[k].c, 0.0f, 0, 0.0f);
                    if (LATE) {   // (ended implies live)
                        io.st_robot(5 + RS * k, r[k].x, r[k].y, r[k].th, 0.0f, 0.0f, 0.0f);
#pragma unroll
                        for (int i = 0; i < 5; ++i) io.st(S, 5 + RS * k + 6 + i, 0.0f);
                    }
                }
                ball = Body{};
                ball.x = np[N].x; ball.y = np[N].y;
                write_obs_nb<KIND, TASK>(P, ob, N, NBLUE, false, true, ball.x, ball.y, ball.vx, ball.vy, 0.0f, 0.0f, 0.0f, 0, 0.0f);
            }
        }
        // ---- observation out: this lane's row ----
        if (live) epl_store_row<OD>(bufs.obs, eo, ob);
    }

    RSX_STAMP(11);
    // ---- store (wire format: degrees, deg/s, infrared, wheel speeds) ----
    if (live) {
#pragma unroll
        for (int k = 0; k < (LATE ? 0 : N); ++k) {   // (the lean form stored them robot by robot above)
            io.st_robot(5 + RS * k, r[k].x, r[k].y, r[k].th, r[k].vx, r[k].vy, wdeg[k]);
            io.st(S, 5 + RS * k + 6, r[k].ir ? 1.0f : 0.0f);
#pragma unroll
            for (int i = 0; i < 4; ++i) io.st(S, 5 + RS * k + 7 + i, wheels[k][i]);
        }
        io.st(S, 0, ball.x); io.st(S, 1, ball.y); io.st(S, 2, K::r_ball + ball.z); io.st(S, 3, ball.vx); io.st(S, 4, ball.vy);
        io.st(S, P.state_dim, ball.vz);
        io.st(S, P.state_dim + 1, ball.om);
        io.st(A, ROW_STEPS, __int_as_float(steps));
        io.st(A, ROW_EPISODE, __uint_as_float(episode));
        io.st(A, ROW_EP_RET, ep_ret);
        if (HAS_TS) io.st(A, ROW_PREV_POT, prev_pot);
    }
    if (LATE) {
        if (blockIdx.x == 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
            bufs.metrics[0] = bufs.metrics[0] + (unsigned long long)P.num_envs;
    } else if (counts_steps) bufs.metrics[0] = steps_before + (unsigned long long)P.num_envs * (unsigned long long)n_steps;
    RSX_STAMP(12);
#ifdef RSX_TIMING
    __builtin_amdgcn_s_waitcnt(0x0F70);
    RSX_STAMP(13);
#endif
}

}  // namespace rsx
