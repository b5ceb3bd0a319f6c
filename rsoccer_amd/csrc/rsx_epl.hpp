// rsx_epl.hpp — VSS-v0 3v3 fused step, "one lane per env" layout for LARGE batches.
//
// Same path, same buffers, same arithmetic as task_step_kernel<VSS, 8, VSS_V0, 6, ...> in
// rsx_kernels.hpp (reference: vss_gym.py:93-311 around robosim.VSS.step, rsim.py:102,105), other
// mapping: lane = env, a wave owns 64 envs and walks the 7 bodies of each one after the other with
// the whole env in registers.  What it buys at scale:
//   * no role divergence and no idle slot: every lane runs "robot k" (or the ball, or the reward)
//     at the same time, so a wave issues each instruction for 64 envs instead of 8 — about half
//     the VALU work per env-step of the 8-lanes-per-env layout, which is VALU-bound there;
//   * a pair of bodies is tested once (21 tests per sub-step instead of 49 lane-pair tests);
//   * every SoA row access of a wave is 256 contiguous bytes.
// What it costs: a wave's critical path is seven times longer, so the layout only pays when
// there are enough envs to fill the chip with such waves (>= ~64 k; the host picks it by batch
// size, RSX_LAYOUT=lanes|epl overrides).  Results are bit-identical to the other layout: every
// body's contact sum is accumulated in partner-index order, each side of a pair evaluates the
// response from its own point of view with the same expressions, draws use the same Philox
// counters (tests/test_gpu_parity.py::test_env_per_lane_layout_is_bit_identical).
#pragma once
#include "rsx_epl_common.hpp"



namespace rsx {

constexpr int EPL_NR = 6;           // VSS 3v3
constexpr int EPL_NB = EPL_NR + 1;  // + ball (body index EPL_NR)
constexpr int EPL_OD = 40;

struct EplShared {
    // contact sums of a sub-step (only when some lane touches something), column = lane; the same words serve
    // as scratch for the poses of a reset placement.  Observations do NOT pass through LDS: each lane
    // holds its env's 40 values in registers and stores them as ten 16-byte pieces (the index arithmetic
    // of a staged, coalesced copy-out cost 5 % of the kernel's VALU instructions and a wave of occupancy).
    EplSums<EPL_NB> c;
};

// VSS-v0 observation of a 3v3 env into registers (vss_gym.py:93-117): same values as write_obs<VSS, VSS_V0>
// with the robot counts known at compile time (a register array cannot be indexed by run-time offsets)
__device__ __forceinline__ void epl_obs_ball(const Params& P, float* ob, float x, float y, float vx, float vy) {
    using T = TC<RSX_TASK_VSS_V0>;
    ob[0] = clampf(x * P.inv_max_pos, -1.2f, 1.2f); ob[1] = clampf(y * P.inv_max_pos, -1.2f, 1.2f);
    ob[2] = clampf(vx * T::inv_max_v, -1.2f, 1.2f); ob[3] = clampf(vy * T::inv_max_v, -1.2f, 1.2f);
}
__device__ __forceinline__ void epl_obs_robot(const Params& P, float* ob, const int k /* constant after unrolling */, float x,
                                              float y, float vx, float vy, float sn, float cs, float om_deg) {
    using T = TC<RSX_TASK_VSS_V0>;
    if (k < 3) {   // blue: 7 values
        float* r = ob + 4 + 7 * k;
        r[0] = clampf(x * P.inv_max_pos, -1.2f, 1.2f); r[1] = clampf(y * P.inv_max_pos, -1.2f, 1.2f);
        r[2] = sn; r[3] = cs;
        r[4] = clampf(vx * T::inv_max_v, -1.2f, 1.2f); r[5] = clampf(vy * T::inv_max_v, -1.2f, 1.2f);
        r[6] = clampf(om_deg * T::inv_max_w, -1.2f, 1.2f);
    } else {        // yellow: 5 values
        float* r = ob + 4 + 7 * 3 + 5 * (k - 3);
        r[0] = clampf(x * P.inv_max_pos, -1.2f, 1.2f); r[1] = clampf(y * P.inv_max_pos, -1.2f, 1.2f);
        r[2] = clampf(vx * T::inv_max_v, -1.2f, 1.2f); r[3] = clampf(vy * T::inv_max_v, -1.2f, 1.2f);
        r[4] = clampf(om_deg * T::inv_max_w, -1.2f, 1.2f);
    }
}
template <int MODE>
__device__ __forceinline__ void vss_epl_body(RSX_HOT_ARGS, const Params& P_, const Buffers& bufs_) {
    constexpr int KIND = RSX_KIND_VSS, TASK = RSX_TASK_VSS_V0, N = EPL_NR;
    using K = KC<KIND>;
    using T = TC<TASK>;
    constexpr int ID = T::info_dim;
    Params P = P_; RSX_UNPACK_HOT(P);
    Buffers bufs = bufs_; bufs.state = hp_state; bufs.aux = hp_aux; bufs.actions = hp_in; bufs.flags = hp_flags;
    const int n_steps = MODE == MODE_ROLLOUT ? (hp_n_steps & RSX_N_STEPS_MASK) : 1;
    __shared__ EplShared sh;
    const int lane = threadIdx.x;
    // step counter of this launch (rsx_kernels.hpp: step_tick).  Device-keyed launches also pick the tile direction here
    // (the host cannot: it does not know the tick's parity), host-keyed ones get it as the sign of hp_per_xcd
    const bool tick_dev = (hp_n_steps & RSX_TICK_DEV) != 0;
    const StepTick tk = step_tick(tick_dev, P, bufs, (uint32_t)n_steps);
    if (__builtin_expect(!tk.ok, 0)) return;
    const int tile = tile_of_block_zigzag(zigzag_per(tick_dev, tk.t, hp_per_xcd));
    const int e_raw = tile * 64 + lane;
    const bool live = e_raw < P.num_envs;
    // lanes beyond the batch shadow its last env: every load is valid and unconditional (no branch per load),
    // stores and counters are masked
    const int e = live ? e_raw : P.num_envs - 1;
    const size_t B = (size_t)P.num_envs;
    const uint32_t env_id = P.env_id_base + (uint32_t)e;   // (the reset path derives its own copy from eo)
    const EplIO io(bufs.state, bufs.aux, P.row_stride, e);   // this lane's column of the [rows][B] arrays (rsx_epl_common.hpp)
    const uint32_t eo = io.eo;
    const __amdgpu_buffer_rsrc_t S = io.S, A = io.A;

    // Register budget: four waves per SIMD need <= 128 VGPRs, so in single-step launches nothing is
    // kept in registers longer than it is needed: the OU state goes back to memory as soon as the
    // commands exist, the cumulative info terms are fetched after the physics, the state rows are
    // stored robot by robot while the observation is assembled.  Multi-step launches (MODE_ROLLOUT)
    // keep all of it in registers across steps instead.
    constexpr bool STEP = MODE == MODE_STEP;
    // ---- load: every row access of the wave is 256 contiguous bytes ----
    Body r[N];
    Body ball = Body{};
    float ou[N][2];
    float wdeg[N];                        // multi-step launches: the wire-format yaw rate (deg/s) of the last step
    float info[6] = {0, 0, 0, 0, 0, 0};   // rows 0, 4, 5 (goal counters) are zero except on a terminal step
    float prev_pot = 0.0f;
    int steps = 0; uint32_t episode = 0;
    float raw[N][6], rawb[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int f = 0; f < 6; ++f) raw[k][f] = io.ld(S, 5 + 6 * k + f);
        ou[k][0] = ou[k][1] = 0.0f;
        if (k >= 1) { ou[k][0] = io.ld(A, ROW_OU + 2 * k); ou[k][1] = io.ld(A, ROW_OU + 2 * k + 1); }
    }
    {
#pragma unroll
        for (int f = 0; f < 5; ++f) rawb[f] = io.ld(S, f);
        rawb[5] = io.ld(S, P.state_dim);
        rawb[6] = io.ld(S, P.state_dim + 1);
        if (!STEP) {   // (single-step launches fetch the episode bookkeeping after the physics: nothing of it is live before)
            steps = __float_as_int(io.ld(A, ROW_STEPS));
            episode = __float_as_uint(io.ld(A, ROW_EPISODE));
#pragma unroll
            for (int i = 1; i <= 3; ++i) info[i] = io.ld(A, ROW_INFO + i);
        }
    }
    const bool counts_steps = blockIdx.x == 0 && lane == 0;   // metrics[0]: see task_step_kernel (read-modify-write at the end: no 64-bit value held across the step)
    const bool fed = MODE == MODE_STEP && bufs.actions != nullptr;
    float act0 = 0.0f, act1 = 0.0f;
    if (fed) {
        const __amdgpu_buffer_rsrc_t AC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bufs.actions), 0, -1, 0x00020000);
        act0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(AC, (int)(2u * eo), 0, 0));
        act1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(AC, (int)(2u * eo), 4, 0));
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): all loads land once, before the step loop
#pragma unroll
    for (int k = 0; k < N; ++k) {   // interpret_body, robot
        r[k] = Body{};
        r[k].x = raw[k][0]; r[k].y = raw[k][1]; r[k].vx = raw[k][3]; r[k].vy = raw[k][4];
        r[k].th = raw[k][2];
        wdeg[k] = raw[k][5];
        r[k].om = raw[k][5] * K::deg2rad;
        sincos_f32(r[k].th * K::deg2rad, r[k].s, r[k].c);
        if (STEP) __builtin_amdgcn_sched_barrier(0);
    }
    ball.x = rawb[0]; ball.y = rawb[1]; ball.vx = rawb[3]; ball.vy = rawb[4];
    ball.z = rawb[2] - K::r_ball; ball.vz = rawb[5]; ball.om = rawb[6];
    // the three internal ball rows (height, vertical speed, spin) rarely change in VSS: they are
    // written back only when they did
    int ball_extra_flag = (rawb[2] != K::r_ball || rawb[5] != 0.0f || rawb[6] != 0.0f) ? 1 : 0;   // anything but "resting, no spin"
    asm volatile("" : "+v"(ball_extra_flag));   // decided HERE: one flag across the step instead of the three rows it is made of
    const bool ball_extra_in = ball_extra_flag != 0;
    bool new_episode = false;

    float reward = 0.0f; int term = 0, trunc = 0;

    for (int it = 0; it < n_steps; ++it) {
        const uint32_t t = tk.t + (uint32_t)it;   // see task_step_kernel
        if (STEP || it == 0) {
            // The previous ball potential (vss_gym.py:256-283) is the potential of the ball where this step
            // finds it: the same expression on the same floats as last step's, so the same number as the one
            // the 8-lane kernel keeps in memory — 8 bytes per env-step less traffic for ~40 instructions
            // (later steps of a multi-step launch carry it in a register)
            const float bx = ball.x, by = ball.y;
            float dx_d = (P.hl_goal + bx) * 100.0f, dx_a = (P.hl_goal - bx) * 100.0f, dy = by * 100.0f;
            float dy2 = 2.0f * (dy * dy);
            float dist_1 = -sqrtf(dx_a * dx_a + dy2), dist_2 = sqrtf(dx_d * dx_d + dy2);
            prev_pot = ((dist_1 + dist_2) * P.inv_len_cm - 1.0f) * 0.5f;
            if (STEP) asm volatile("" : "+v"(prev_pot));   // computed HERE: one value across the physics, not the two coordinates it is made of
        }
        // ---- actions -> commands (vss_gym.py:119-142,235-254) ----
        float q0[N], q1[N];
        u32x4 blk = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float a0, a1;
            if ((k & 1) == 0) blk = philox4x32(env_id, 0u, t, DOM_ACT | ((uint32_t)(k >> 1) << 8), P.key0, P.key1);   // one block serves two robots
            const uint32_t w0 = (k & 1) ? blk.z : blk.x, w1 = (k & 1) ? blk.w : blk.y;
            if (k == 0) {
                if (fed) { a0 = act0; a1 = act1; }
                else { a0 = u01(w0) * 2.0f - 1.0f; a1 = u01(w1) * 2.0f - 1.0f; }
            } else {  // Ornstein-Uhlenbeck noise, Utils/Utils.py:14-21 (Box-Muller on Philox)
                float u1 = (float)((w0 >> 8) + 1u) * 5.9604644775390625e-08f;
                float ang = (u01(w1) - 0.5f) * 6.283185307179586f;
                float rad = sqrtf(-2.0f * log_f32(u1));
                float sn, cs;
                sincos_f32(ang, sn, cs);
                float n0 = rad * cs, n1 = rad * sn;
                ou[k][0] = (ou[k][0] + P.ou_theta_dt * (0.0f - ou[k][0])) + P.ou_sig_sqdt * n0;
                ou[k][1] = (ou[k][1] + P.ou_theta_dt * (0.0f - ou[k][1])) + P.ou_sig_sqdt * n1;
                a0 = ou[k][0]; a1 = ou[k][1];
            }
            q0[k] = vss_wheel(a0); q1[k] = vss_wheel(a1);
            const float qq[2] = {q0[k], q1[k]};
            robot_targets<KIND>(P, r[k], qq);
            if (STEP && k >= 1 && live) { io.st(A, ROW_OU + 2 * k, ou[k][0]); io.st(A, ROW_OU + 2 * k + 1, ou[k][1]); }
            if (STEP) __builtin_amdgcn_sched_barrier(0);   // one robot after the other: interleaving them for ILP costs a wave of occupancy
        }
        float energy = -(fabsf(q0[0]) + fabsf(q1[0]));   // the agent's wheel commands: energy term of the reward
        asm volatile("" : "+v"(energy));   // computed HERE: one value across the physics instead of the two commands

        // ---- physics: n_sub sub-steps, the whole env in registers ----
        ball_step_friction(P, ball);   // rolling resistance + spin decay, once per step() (rsx_body.hpp, like every per-body formula below)
        for (int sub = 0; sub < P.n_sub; ++sub) {
            // A: actuation + integration
#pragma unroll
            for (int k = 0; k < N; ++k) integrate_robot<KIND>(P, r[k]);
            integrate_ball<KIND>(P, ball);

            // B: contacts, Jacobi over the post-integration snapshot.  Every pair once; the exact
            // compare d2 < thr gives one bit per touching pair (0 < d2: see find_touching).
            // A second sweep over the corrected snapshot runs in the envs where anything touched.
            const bool ball_low = ball.z < K::robot_h;
            // bits of `touching` that involve body k (pairs in lexicographic order, see epl_pair)
            constexpr unsigned PM[EPL_NB] = {epl_pair_mask<EPL_NB>(0), epl_pair_mask<EPL_NB>(1), epl_pair_mask<EPL_NB>(2), epl_pair_mask<EPL_NB>(3),
                                             epl_pair_mask<EPL_NB>(4), epl_pair_mask<EPL_NB>(5), epl_pair_mask<EPL_NB>(6)};
            auto find_touching = [&]() -> unsigned {
                // bit p = pair p touches.  The pairs are visited in REVERSE order and each result is shifted in from the
                // right (acc = acc + acc + bit: a compare and ONE add-with-carry per pair, no bit constant in a register),
                // so pair p ends up at bit p; the ball's height gates its six pairs once, at the end
                unsigned touching = 0;
#pragma unroll
                for (int i = N - 1; i >= 0; --i) {
#pragma unroll
                    for (int j = N; j > i; --j) {
                        const float xj = j == N ? ball.x : r[j < N ? j : 0].x, yj = j == N ? ball.y : r[j < N ? j : 0].y;
                        const float dx = xj - r[i].x, dy = yj - r[i].y;
                        // one compare per pair: d2 < thr.  The model's 0 < d2 (bodies in one place have no normal and are not a contact) is
                        // applied where a pair is walked (epl_walk_pairs) — `bits(d2) - 1 < bits(thr) - 1` did both here, for one
                        // more instruction on each of the 21 pairs of every sub-step.  (Spelled as the two instructions it is: the
                        // compiler's own choice for `2 acc + (d2 < thr)` is a compare, a select of the bit and a share of a shift and an OR.)
                        const float d2 = fma_(dx, dx, dy * dy);
                        asm("v_cmp_gt_f32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(touching) : "v"(d2), "s"(j == N ? K::rs_rb2 : K::rs_rr2) : "vcc");
                    }
                }
                return ball_low ? touching : (touching & ~PM[N]);
            };
            bool deep = false, wallp = false;
            for (int sweep = 0; sweep < 4; ++sweep) {   // the later sweeps run the same (cached) instructions
                // second: envs with a deep pair only; third and fourth: envs whose last sweep also saw a wall pair (model v2)
                const bool mine = sweep == 0 || (deep && (sweep == 1 || wallp));
                if (sweep >= 1 && !__any(mine)) break;   // no env of the wave goes on: no further pair test either
                unsigned touching = mine ? find_touching() : 0u;
                if (!__any(touching != 0)) break;
                // some env of the wave has a contact: sums in LDS, pairs walked per lane, both sides of a pair from one normal
                // (epl_walk_pairs, rsx_epl_common.hpp: the ball is body N, a circle like the robots)
                epl_zero_sums(sh.c, lane);
                wave_sync();
                deep = false; wallp = false;
                unsigned dropped = 0u;   // pairs at zero distance: not a contact, and their bodies not "touched" by them
                epl_walk_pairs<KIND, N, true, true>(P, r, ball, sh.c, lane, touching, deep, wallp, &dropped);
                touching &= ~dropped;
                wave_sync();
                epl_apply_sums<N>(r, ball, sh.c, lane, [&](int k) { return (touching & PM[k]) != 0; });   // (the registers still held the snapshot)
                wave_sync();
            }
            // C: walls
#pragma unroll
            for (int k = 0; k < N; ++k) robot_walls<KIND>(P, r[k]);
            ball_walls<KIND>(P, ball);
        }

        // ---- wire-format values, observation, reward ----
        if (STEP) {   // episode bookkeeping and cumulative shaping terms: fetched now, used after the observation
            steps = __float_as_int(io.ld(A, ROW_STEPS));
            episode = __float_as_uint(io.ld(A, ROW_EPISODE));
#pragma unroll
            for (int i = 1; i <= 3; ++i) info[i] = io.ld(A, ROW_INFO + i);
        }
        const bool first_step = steps == 0;
        float ob[EPL_OD];   // this env's observation, in registers
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float wd = r[k].om * K::rad2deg;
            wdeg[k] = wd;
            r[k].om = wd * K::deg2rad;
            sincos_f32(r[k].th * K::deg2rad, r[k].s, r[k].c);
            epl_obs_robot(P, ob, k, r[k].x, r[k].y, r[k].vx, r[k].vy, r[k].s, r[k].c, wd);
            if (STEP && live) {   // wire format, robot by robot (an env that resets below writes its rows again)
                io.st_robot(5 + 6 * k, r[k].x, r[k].y, r[k].th, r[k].vx, r[k].vy, wd);
            }
            if (STEP) __builtin_amdgcn_sched_barrier(0);
        }
        ball.z = (K::r_ball + ball.z) - K::r_ball;
        epl_obs_ball(P, ob, ball.x, ball.y, ball.vx, ball.vy);
        if (first_step) { info[1] = 0.0f; info[2] = 0.0f; info[3] = 0.0f; }
        info[0] = 0.0f; info[4] = 0.0f; info[5] = 0.0f;
        {   // vss_gym.py:144-192,256-311
            reward = 0.0f; term = 0;
            const float bx = ball.x, by = ball.y;
            if (bx > P.half_len) { info[0] += 1.0f; info[4] += 1.0f; reward = 10.0f; term = 1; }
            else if (bx < -P.half_len) { info[0] -= 1.0f; info[5] += 1.0f; reward = -10.0f; term = 1; }
            else {
                float dx_d = (P.hl_goal + bx) * 100.0f, dx_a = (P.hl_goal - bx) * 100.0f, dy = by * 100.0f;
                float dy2 = 2.0f * (dy * dy);
                float dist_1 = -sqrtf(dx_a * dx_a + dy2), dist_2 = sqrtf(dx_d * dx_d + dy2);
                float pot = ((dist_1 + dist_2) * P.inv_len_cm - 1.0f) * 0.5f;
                float grad = 0.0f;
                if (!first_step) grad = clampf((pot - prev_pot) * 3.0f * P.inv_dt, -5.0f, 5.0f);
                prev_pot = pot;
                float rbx = bx - r[0].x, rby = by - r[0].y;
                float nrm = sqrtf(rbx * rbx + rby * rby);
                float mv = nrm > 0.0f ? (rbx / nrm) * r[0].vx + (rby / nrm) * r[0].vy : 0.0f;   // unguarded in vss_gym.py:298
                float move = clampf(mv * 2.5f, -5.0f, 5.0f);
                float t_move = 0.2f * move, t_grad = 0.8f * grad, t_en = 2e-4f * energy;
                reward = (t_move + t_grad) + t_en;
                info[1] += t_move; info[2] += t_grad; info[3] += t_en;
            }
        }
        steps += 1;
        trunc = steps >= P.max_steps;
        const bool ended = live && (term | trunc);
        if (live) {
#pragma unroll
            for (int i = 1; i <= 3; ++i) io.st(A, ROW_INFO + i, info[i]);
            if (term || first_step) {   // goal counters: non-zero on a terminal step only, cleared on the next first step
                io.st(A, ROW_INFO + 0, info[0]); io.st(A, ROW_INFO + 4, info[4]); io.st(A, ROW_INFO + 5, info[5]);
            }
            io.st(A, ROW_REWARD, reward);
            io.st_flags(bufs.flags, P.num_envs, term, trunc);
        }

        // ---- episode end: same-step auto-reset, one lane = one env ----
        if (__any(ended)) {
            if (ended) {
                epl_store_row<EPL_OD>(bufs.final_obs, eo, ob);   // terminal observation
                episode += 1; new_episode = true;
                unsigned long long* const ms = metric_slot(bufs);
                atomicAdd(&ms[1], 1ull);
                if (info[4] > 0.0f) atomicAdd(&ms[2], 1ull);
                if (info[5] > 0.0f) atomicAdd(&ms[3], 1ull);
                atomicAdd(&ms[4], (unsigned long long)__float2ll_rn(vss_episode_return(info) * 1048576.0f));
                atomicAdd(&ms[5], (unsigned long long)steps);
                if (trunc && !term) atomicAdd(&ms[6], 1ull);
                // placement: the reference's sequential rejection sampling (vss_gym.py:194-233)
                uint32_t n = 0;
                uint32_t eo_r = eo;
                asm volatile("" : "+v"(eo_r));   // an opaque copy: otherwise the first Philox round of the step's draws (same counter word) is kept alive across the physics for this rare path
                const uint32_t env_id_r = P.env_id_base + (eo_r >> 2);
                auto draw = [&]() -> float2 {
                    const u32x4 u = philox4x32(env_id_r, episode, n++, DOM_PLACE, P.key0, P.key1);
                    return make_float2(u01(u.x), u01(u.y));
                };
                float bx, by;
                { const float2 u = draw(); bx = P.pl_xlo + P.pl_xspan * u.x; by = P.pl_ylo + P.pl_yspan * u.y; }
                // scratch for the poses placed so far: this lane's column of the (now idle) contact sums
                const int lane_r = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // = lane (64-thread workgroups), re-derived: not kept across the physics for this rare path
                float* const px = &sh.c.acc[0][0][lane_r], * const py = &sh.c.acc[1][0][lane_r], * const pth = &sh.c.acc[2][0][lane_r];
                for (int k = 0; k < N; ++k) {
                    float x = 0.0f, y = 0.0f;
                    for (int tt = 0; tt < 64; ++tt) {
                        const float2 u = draw();
                        x = P.pl_xlo + P.pl_xspan * u.x;
                        y = P.pl_ylo + P.pl_yspan * u.y;
                        bool ok = true;
                        { float dx = x - bx, dy = y - by; if (dx * dx + dy * dy < P.pl_min_d2) ok = false; }
                        for (int q = 0; q < k; ++q) {
                            float dx = x - px[q * 64], dy = y - py[q * 64];
                            if (dx * dx + dy * dy < P.pl_min_d2) ok = false;
                        }
                        if (ok) break;
                    }
                    const float2 u = draw();
                    px[k * 64] = x; py[k * 64] = y; pth[k * 64] = 360.0f * u.x;
                }
                steps = 0;   // VSS-v0 keeps prev_pot (the first step of an episode ignores it)
                float nx[N], ny[N], nth[N];
#pragma unroll
                for (int k = 0; k < N; ++k) { nx[k] = px[k * 64]; ny[k] = py[k * 64]; nth[k] = pth[k * 64]; }
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    ou[k][0] = 0.0f; ou[k][1] = 0.0f;
                    r[k] = Body{};
                    r[k].x = nx[k]; r[k].y = ny[k];
                    r[k].th = nth[k];
                    wdeg[k] = 0.0f;
                    sincos_f32(r[k].th * K::deg2rad, r[k].s, r[k].c);
                    epl_obs_robot(P, ob, k, r[k].x, r[k].y, r[k].vx, r[k].vy, r[k].s, r[k].c, 0.0f);
                    if (STEP && live) {   // this env's rows were written before the reset was known
                        io.st_robot(5 + 6 * k, r[k].x, r[k].y, r[k].th, 0.0f, 0.0f, 0.0f);
                        if (k >= 1) { io.st(A, ROW_OU + 2 * k, 0.0f); io.st(A, ROW_OU + 2 * k + 1, 0.0f); }
                    }
                }
                ball = Body{};
                ball.x = bx; ball.y = by;
                epl_obs_ball(P, ob, ball.x, ball.y, ball.vx, ball.vy);
            }
        }
        // ---- observation out: this lane's row, ten 16-byte stores ----
        if (live) epl_store_row<EPL_OD>(bufs.obs, eo, ob);
    }

    // ---- store (wire format) ----
    if (live) {
        if (!STEP) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                io.st_robot(5 + 6 * k, r[k].x, r[k].y, r[k].th, r[k].vx, r[k].vy, wdeg[k]);
                if (k >= 1) { io.st(A, ROW_OU + 2 * k, ou[k][0]); io.st(A, ROW_OU + 2 * k + 1, ou[k][1]); }
            }
        }
        io.st(S, 0, ball.x); io.st(S, 1, ball.y); io.st(S, 3, ball.vx); io.st(S, 4, ball.vy);
        const float z_out = K::r_ball + ball.z;
        if (ball_extra_in || z_out != K::r_ball || ball.vz != 0.0f || ball.om != 0.0f) {   // was or is off its resting values
            io.st(S, 2, z_out); io.st(S, P.state_dim, ball.vz); io.st(S, P.state_dim + 1, ball.om);
        }
        io.st(A, ROW_STEPS, __int_as_float(steps));
        if (!STEP || new_episode) io.st(A, ROW_EPISODE, __uint_as_float(episode));
    }
    if (counts_steps) bufs.metrics[0] = bufs.metrics[0] + (unsigned long long)P.num_envs * (unsigned long long)n_steps;
}

// Two entry points, two register budgets (measured).  Single-step launches are limited by the
// memory system: 154 VGPRs without a spill (3 waves per SIMD) beat 128 VGPRs with 84 B of scratch per lane —
// the spills alone were 150 MB of the 794 MB a 1 M-env launch moved; without them it moves 631 MB, 1.11 x the
// algorithmic bytes.  Multi-step launches are limited by instruction issue and prefer the fourth wave.
#ifndef RSX_EPL_WAVES
#define RSX_EPL_WAVES 4
#endif
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RSX_EPL_WAVES, RSX_EPL_WAVES))) void vss_epl_kernel(RSX_HOT_ARGS, const Params P_, const Buffers bufs_) {
    static_assert(MODE == MODE_STEP, "single-step entry point");
    vss_epl_body<MODE_STEP>(hp_state, hp_aux, hp_in, hp_flags, hp_num_envs, hp_state_dim, hp_per_xcd, hp_n_steps, P_, bufs_);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void vss_epl_rollout_kernel(RSX_HOT_ARGS, const Params P_, const Buffers bufs_) {
    vss_epl_body<MODE_ROLLOUT>(hp_state, hp_aux, hp_in, hp_flags, hp_num_envs, hp_state_dim, hp_per_xcd, hp_n_steps, P_, bufs_);
}

}  // namespace rsx
