// rsx_kernels.hpp — the per-env.step() hot path as hand-written HIP for gfx950 (CDNA4).
//
// Replaces robosim.VSS.step / robosim.SSL.step (+ get_state) — reference call sites
// rsoccer_gym/Simulators/rsim.py:102,105,155,158 — and, in the fused task variants, the Python
// hooks around them: VSSEnv._get_commands/_frame_to_observations/_calculate_reward_and_done
// (vss/env_vss/vss_gym.py:93-192) and the same three of SSLHWStaticDefendersEnv
// (ssl/ssl_hw_challenge/static_defenders.py:90-212), TimeLimit and reset placement.
//
// Mapping (wave64):
//   * one lane owns one rigid body (robot or ball); an env occupies a group of L lanes
//     (L = 8 for <= 7 robots, 16, 32, or 64 = the whole wavefront) and a wave hosts G = 64/L
//     envs.  L >= 16: lane = body * G + env_in_wave, so the G lanes that own "body k" of
//     neighbouring envs are adjacent and read adjacent floats of SoA row k (G*4 contiguous bytes
//     per body).  L = 8: lane = env_in_wave * 8 + body (LaneMap below; measured -1 % at the
//     latency-bound batches, +2-3 % at 65 536 envs).  Either way a wave touches the same 32-byte
//     pieces, and the 6..11 rows of one body share cache lines with the neighbouring tiles handled
//     by the SAME XCD (tile -> XCD map below), so every byte fetched into an L2 is used.
//   * the all-pairs contact test is a Jacobi sweep: each lane publishes (x, y, vx, vy) as one
//     float4 in LDS and reads the other bodies' float4 back (ds_read_b128, broadcast inside a
//     group, G distinct 16-B slots per instruction -> conflict free).  With the robot count a
//     template constant the sweep is fully unrolled, so all reads of a sub-step are in flight
//     together and their latency is paid once.  All sub-steps of one env.step() run out of
//     registers + LDS; HBM is touched once for the load and once for the store of the state.
//   * SSL: the robot lane evaluates its own robot-ball contact (kicker-mouth geometry) and
//     publishes the ball-side impulse / dribbler / kick record; the ball lane learns from one
//     ballot which robots wrote one and sums those in index order.  No lane re-derives another
//     lane's geometry.
//   * a single-step launch lasts as long as its slowest wave, so the rare paths are shaped for
//     the wave that takes them: the overlap test of a sweep is one integer minimum and one
//     compare, each lane then walks only ITS partners (lanes with different partners share an
//     iteration), and the reset placement of an ended env is done by all of its lanes together
//     (place_env_parallel).
//   * 64-thread workgroups (one wave): a 4096-env VSS batch is 512 workgroups, two per CU,
//     so all 256 CUs work; block b runs on XCD b % 8 and is mapped to tile
//     (b % 8) * tiles_per_xcd + b / 8 so each XCD's L2 sees one contiguous 1/8 of every row.
//   * fp32 throughout, -ffp-contract=off, fixed summation order (body index order) -> results
//     are bit-identical for any batch size, position in the batch, L, or shard.
//   * scalar registers: class/task constants are instruction literals (KC<>, TC<>); only the
//     run-time block `Params` and 8 base pointers live in SGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "rsx_math.hpp"
#include "rsx_params.hpp"
#include "rsx_body.hpp"

namespace rsx {

// rows of the per-env scalar arena `aux` ([rows][B], 4 bytes each)
constexpr int ROW_REWARD = 0, ROW_PREV_POT = 1, ROW_EP_RET = 2, ROW_STEPS = 3, ROW_EPISODE = 4,
              ROW_INFO = 5 /* 10 rows */, ROW_OU = 15 /* 2*N rows */;
// ROW_PREV_POT is the per-episode task scalar: previous ball potential (VSS-v0), checkpoint
// counter (dribbling), stalled-step counter (pass endurance)
__host__ __device__ constexpr int aux_rows(int n_robots) { return ROW_OU + 2 * n_robots; }

struct Buffers {
    float* state;          // [state_dim+X_ROWS][B]
    float* aux;            // [aux_rows][B]   reward, prev_pot, ep_ret, steps, episode, info, ou
    float* obs;            // [B][obs_dim]
    float* final_obs;      // [B][obs_dim]
    uint8_t* flags;        // [3][B]          terminated, truncated, env mask of reset_to (MODE_REFRESH)
    const float* cmds;     // [N*C][B]        (raw simulator path)
    const float* actions;  // [B][act_dim] or nullptr = random
    unsigned long long* metrics;  // [RSX_METRICS]
    unsigned long long* mslots;   // [MSLOTS][RSX_METRICS]: per-block-group partial sums of the episode counters (see metric_slot)
    float* pcache;                // placement cache (see placement_helper): [2][3 * (N + 1) + 1][B], or nullptr
    unsigned long long* pcstats;  // [2] resets served from the cache / placed inline (nullptr unless RSX_PCACHE_STATS=1)
#ifdef RSX_TIMING
    unsigned long long* dbg;      // [8][gridDim] s_memtime stamps (development builds only)
#endif
};

// (Observation values go straight from the lane that owns them to the row in HBM — scattered 4-byte stores inside the tile's contiguous
// run of rows — not through an LDS staging area + a coalesced copy-out: measured, VSS-v0 4096 envs 10.01 -> 9.74 us per step,
// 65 536 envs 26.1 -> 25.2; the SSL tasks 0-1 %.)
template <int L>
struct Shared {
    float4 A[64];   // x, y, vx, vy of every body (slot = lane)
    float4 Bq[64];  // SSL robot -> ball record 0: dvx, dvy, dpx, dpy (ball side)
    float4 Cq[64];  // SSL robot -> ball record 1: flags, ovx, ovy, ovz
    float Dq[64];   // SSL robot -> ball record 2: spin change of the ball
    float W[64];    // robots: yaw rate, ball: spin (rad/s) — read on the contact path only
    float2 F[64];   // VSS: held axes of the body in this sweep's snapshot (rsx_body.hpp: held_axes) — read on the contact path only
    alignas(16) float X[64], Y[64];   // positions once more, [env slot][body]: four partners per 16-byte read for the packed overlap test
    float x0[64 / L][12];      // robot 0 -> reward lane exchange
    float2 draws[64 / L][L < 16 ? 16 : L];  // placement: speculative Philox draws of an ended env
    uint32_t ep[64];           // placement helper: the episode ids of the wave's 64 envs
#ifdef RSX_TIMING
    unsigned long long* dbg;   // development builds: where the sub-step stamps go (nullptr = none)
#endif
};

// Marks a branch as seldom taken so that its body is laid out away from the hot path.  Which hints
// pay off was measured per simulator class (single-step launch): SSL takes all three (static
// defenders 10.7 -> 10.5 us, pass endurance 11.2 -> 10.85); VSS takes the contact sweep (2) and
// the episode end (4) but not the airborne-ball test (1): 8.55 -> 8.47 us (all three: 8.62).
#define RSX_RARE_B(KIND, bit, c) (((KIND) == RSX_KIND_SSL || (6 & (bit))) ? __builtin_expect(!!(c), 0) : !!(c))

// Addresses into the [rows][B] arrays on the hot paths: a uniform base pointer (scalar registers) + ONE 32-bit BYTE offset per
// lane — the global_load / global_store "saddr" form, no 64-bit vector multiply-adds and shifts per access.  The host refuses
// batches whose arrays would reach 4 GB (rsx_api.hip: RSX_ERR_ARG at create / attach).
typedef uint32_t ix_t;
__device__ __forceinline__ float& at_byte(float* base, const ix_t off) { return *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off); }
__device__ __forceinline__ const float& at_byte(const float* base, const ix_t off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off); }

// Which lane holds body j of the wave's env g (and which LDS slot: slot = lane).  Body-major (lane = j * G + g: the G lanes
// that own "body j" of neighbouring envs are adjacent, every row access is G * 4 contiguous bytes) for L >= 16; env-major
// (lane = g * L + j: an env's eight bodies are eight adjacent lanes) for L == 8 — measured -1 % at 4096 envs, +2-3 % at 65 536.
// Partners are read through the LDS snapshot in every width (reading them through DPP
// row shifts was measured and dropped: profiles/LABBOOK.md).
template <int L>
struct LaneMap {
    static constexpr int G = 64 / L;
    static constexpr bool EM = L == 8;
    static __device__ __forceinline__ int slot(const int j, const int g) { return EM ? g * L + j : j * G + g; }
    static __device__ __forceinline__ int body(const int lane) { return EM ? lane % L : lane / G; }
    static __device__ __forceinline__ int env(const int lane) { return EM ? lane / L : lane % G; }
};
// lanes of the env in slot g
template <int L>
__device__ __forceinline__ unsigned long long env_lane_mask(const int g) {
    constexpr int G = 64 / L;
    if (LaneMap<L>::EM) return ((1ull << L) - 1ull) << (g * L);
    unsigned long long m = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) m |= 1ull << (j * G);
    return m << g;
}

// VSS contact sweep with a run-time partner loop: exact integer overlap test into one bit per
// partner, then the lane walks ITS partners in index order.  First sweep of the run-time-count
// kernels and second sweep (rare) of all VSS kernels.  Returns whether some pair was deep.
template <int KIND, int L>
__device__ __forceinline__ bool vss_sweep_loop(const Params& P, Body& o, const int N, const int g, const bool is_ball,
                                               const bool ball_low, const Shared<L>& sh, bool& wallp, const float2 fo) {
    using K = KC<KIND>;
    constexpr int G = 64 / L;
    constexpr uint32_t T_RR = __builtin_bit_cast(uint32_t, K::rs_rr2) - 1u;
    constexpr uint32_t T_RB = __builtin_bit_cast(uint32_t, K::rs_rb2) - 1u;
    unsigned todo = 0;
#pragma unroll 4
    for (int j = 0; j <= N; ++j) {
        const float4 oj = sh.A[LaneMap<L>::slot(j, g)];
        const bool rb = is_ball || j == N;
        const float dx = oj.x - o.x, dy = oj.y - o.y;
        const uint32_t u = __float_as_uint(fma_(dx, dx, dy * dy)) - 1u;   // own slot: 0xFFFFFFFF
        todo |= ((u < (rb ? T_RB : T_RR)) & (!rb | ball_low)) ? 1u << j : 0u;
    }
    if (todo == 0) return false;
    const bool v2w = K::wall_aware && __ballot(!is_ball && at_wall<KIND>(P, o.x, o.y)) != 0ull;   // (rsx_body.hpp: contact_response)
    bool deep = false;
    float avx = 0.0f, avy = 0.0f, apx = 0.0f, apy = 0.0f, aw = 0.0f;
    const Body snap = o;   // every partner is evaluated against the snapshot
    const float lever = is_ball ? K::r_ball : K::r_robot;
    while (todo) {
        const int j = __builtin_ctz(todo);
        todo &= todo - 1;
        const float4 oj = sh.A[LaneMap<L>::slot(j, g)];
        const float wj = sh.W[LaneMap<L>::slot(j, g)];
        const float2 fj = sh.F[LaneMap<L>::slot(j, g)];
        const float dx = oj.x - o.x, dy = oj.y - o.y;
        const bool rb = is_ball || j == N;
        contact_response<KIND>(P, snap, oj, fma_(dx, dx, dy * dy), rb ? K::rs_rb : K::rs_rr, rb ? K::ope_rb : K::ope_rr,
                         is_ball ? K::w_rb_b : (j == N ? K::w_rb_r : K::w_rr),
                         is_ball ? K::kt_rb_b : (j == N ? K::kt_rb_r : K::kt_rr), rb ? K::mu_rb : K::mu_rr,
                         is_ball ? K::spin_c : 0.0f, fma_(wj, j == N ? K::r_ball : K::r_robot, snap.om * lever),
                         K::beta, K::pen2, !rb, v2w, avx, avy, apx, apy, aw, deep, wallp, fo, fj);
    }
    // only a body that touched something is updated (the others keep their bits)
    o.vx = o.vx + avx; o.vy = o.vy + avy;
    o.x = o.x + apx; o.y = o.y + apy;
    if (is_ball) o.om = o.om + aw;
    return deep;
}

// What the kicker / dribbler of some robot decided for the ball in the first sweep of a sub-step
// u[j] = bits(|p_j - p_o|^2) - 1 for the SLOTS bodies of the lane's env, two partners per packed-FP32 instruction
// (v_pk_add / v_pk_mul / v_pk_fma are IEEE per component: the same bits as the scalar form), positions from the
// [env][body] copies in LDS (one 16-byte read = four partners).
struct NoFill { __device__ __forceinline__ void operator()() const {} };
// `fill`: work that does not depend on the partners' positions, issued between the LDS reads and their first use (the reads take
// ~100 cycles to come back and a lone wave has nothing else to run meanwhile)
template <int SLOTS, int L, typename FILL = NoFill>
__device__ __forceinline__ void overlap_keys_packed(const Shared<L>& sh, const int g, const float ox, const float oy, uint32_t* u, FILL fill = FILL{}) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int Q = (SLOTS + 3) / 4;
    const f4* X4 = reinterpret_cast<const f4*>(&sh.X[g * L]);
    const f4* Y4 = reinterpret_cast<const f4*>(&sh.Y[g * L]);
    f4 xs[Q], ys[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { xs[q] = X4[q]; ys[q] = Y4[q]; }
    fill();   // (in program order behind the reads; a sched_barrier here keeps the compiler from peeling the sweep loop and costs scratch)
    const f2 ox2 = {ox, ox}, oy2 = {oy, oy};
#pragma unroll
    for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int j = 4 * q + 2 * hlf;
            if (j >= SLOTS) continue;
            const f2 px = hlf ? xs[q].zw : xs[q].xy, py = hlf ? ys[q].zw : ys[q].xy;
            const f2 dx = px - ox2, dy = py - oy2;
            const f2 t = dy * dy;
            const f2 d2 = __builtin_elementwise_fma(dx, dx, t);
            u[j] = __float_as_uint(d2.x) - 1u;
            if (j + 1 < SLOTS) u[j + 1] = __float_as_uint(d2.y) - 1u;
        }
    }
}

struct BallOverride { bool ovr, okick; float ovx, ovy, ovz; };

// SSL contact sweep.  Robot lanes: robot-robot pairs (circles), then the robot's own robot-ball
// geometry (kicker mouth or body circle) whose ball-side record goes to LDS; one ballot tells the
// ball lane which robots wrote one.  FIRST: infrared is refreshed and kicker / dribbler act.
// NRX > 0: robot count known at compile time.  `first` is wave-uniform: both sweeps of a sub-step run
// the SAME instructions (a second copy of this code would be cold in the instruction cache every
// time it is needed, which costs more than the sweep itself).
template <int KIND, int L, int NRX>
__device__ __forceinline__ bool ssl_sweep(const Params& P, Body& o, const int N, const int g, const int lane,
                                          const bool is_robot, const bool is_ball, const bool ball_low,
                                          const bool first, Shared<L>& sh, BallOverride& bo, bool& wallp) {
    using K = KC<KIND>;
    constexpr int G = 64 / L;
    constexpr uint32_t T_RR = __builtin_bit_cast(uint32_t, K::rs_rr2) - 1u;
    int fl = 0;   // what this robot does to the ball in this sweep (0 = nothing)
    bool touched = false;   // deep contact seen by this lane
    bool got = false;       // this body touched something: only then is it updated
    float avx = 0.0f, avy = 0.0f, apx = 0.0f, apy = 0.0f, aw = 0.0f;
    if (is_robot) {
        unsigned todo = 0;
        if (NRX) {
            uint32_t u[NRX ? NRX : 1];   // exact integer form of 0 < d2 < rs_rr^2, see the VSS sweep
            overlap_keys_packed<(NRX ? NRX : 1), L>(sh, g, o.x, o.y, u);
            uint32_t um = u[0];
#pragma unroll
            for (int j = 1; j < NRX; ++j) um = min(um, u[j]);
            if (RSX_RARE_B(KIND, 2, um < T_RR)) {
#pragma unroll
                for (int j = NRX - 1; j >= 0; --j)   // slot j ends at bit j: shifted in from the right, highest slot first (a compare and an add-with-carry per slot, no bit constant in a register)
                    asm("v_cmp_gt_u32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(todo) : "v"(u[j]), "s"(T_RR) : "vcc");
            }
        } else {
#pragma unroll 4
            for (int j = 0; j < N; ++j) {
                const float4 oj = sh.A[LaneMap<L>::slot(j, g)];
                const float dx = oj.x - o.x, dy = oj.y - o.y;
                todo |= (__float_as_uint(fma_(dx, dx, dy * dy)) - 1u) < T_RR ? 1u << j : 0u;
            }
        }
        if (RSX_RARE_B(KIND, 2, todo != 0)) {   // per-lane partner walk, see the VSS sweep
            bool& deep = touched;
            got = true;
            const bool v2w = K::wall_aware && __ballot(at_wall<KIND>(P, o.x, o.y)) != 0ull;   // some robot of the wave (that has a partner) at a wall
            // Two copies of the walk, picked by that wave-uniform flag: the usual one holds v1's instructions and nothing else, the wall-
            // aware one (model v2) sits behind it — a test per partner inside ONE loop put the wall code's branches into the hot loop body
            auto walk = [&](auto wall_tag) {
                constexpr bool WALLS = decltype(wall_tag)::value;
                // software-pipelined like the VSS walk: the next partner's slot is fetched while the current response is computed
                int jn = __builtin_ctz(todo);
                todo &= todo - 1;
                float4 nxt = sh.A[LaneMap<L>::slot(jn, g)];
                float nxw = sh.W[LaneMap<L>::slot(jn, g)];
                for (;;) {
                    const float4 oj = nxt;
                    const float wj = nxw;
                    const bool more = todo != 0;
                    if (more) {
                        jn = __builtin_ctz(todo);
                        todo &= todo - 1;
                        nxt = sh.A[LaneMap<L>::slot(jn, g)];
                        nxw = sh.W[LaneMap<L>::slot(jn, g)];
                    }
                    const float dx = oj.x - o.x, dy = oj.y - o.y;
                    contact_response<KIND>(P, o, oj, fma_(dx, dx, dy * dy), K::rs_rr, K::ope_rr, K::w_rr, K::kt_rr, K::mu_rr, 0.0f,
                                           fma_(wj, K::r_robot, o.om * K::r_robot), K::beta, K::pen2, true, L == 8 ? WALLS : v2w, avx, avy, apx, apy, aw, deep, wallp);
                    if (!more) break;
                }
            };
            // (measured, us per step v1 / one loop / two copies: 1v6 at 2048 envs, 8 lanes: 9.28 / 9.82 / 9.48; 11v11 at 1024 envs, 32 lanes:
            // 9.71 / 10.00 / 10.10 — each width keeps its better form)
            if constexpr (L == 8) { if (__builtin_expect(v2w, 0)) walk(std::true_type{}); else walk(std::false_type{}); }
            else walk(std::true_type{});
        }
        // robot - ball: kicker mouth (flat face at dck) or body circle; n points robot -> ball
        const float4 ob = sh.A[LaneMap<L>::slot(N, g)];
        float dx = ob.x - o.x, dy = ob.y - o.y;
        float nx = 0.0f, ny = 0.0f, pen = -1.0f;
        bool mouth = false, touch = false;
        // a mouth, circle or infrared contact needs the ball's centre within 0.126 m of the robot's ((dck_rb + ir_tol)^2
        // + half_kw^2 = 0.126^2, and rs_rb < 0.126): everything farther away skips the geometry (same values when taken)
        constexpr float NEAR2 = 0.13f * 0.13f;
        const float d2 = fma_(dx, dx, dy * dy);
        if (ball_low && d2 < NEAR2) {
            float lx = fma_(dx, o.c, dy * o.s), ly = fma_(dy, o.c, -(dx * o.s));
            if (fabsf(ly) < K::half_kw && lx > 0.0f) {
                mouth = true; pen = K::dck_rb - lx; nx = o.c; ny = o.s; touch = pen > 0.0f;
            } else {
                if (d2 < K::rs_rb2 && d2 > 0.0f) {
                    float d = sqrtf(d2), inv = 1.0f / d;
                    nx = dx * inv; ny = dy * inv; pen = K::rs_rb - d; touch = true;
                }
            }
        }
        float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float dws = 0.0f;
        if (touch) {
            touched |= pen > K::pen2;
            got = true;
            const float dvx = ob.z - o.vx, dvy = ob.w - o.vy;
            float vn = fma_(dvx, nx, dvy * ny);
            if (vn < 0.0f) {
                const float omb = sh.W[LaneMap<L>::slot(N, g)];
                float q = K::ope_rb * vn * K::w_rb_r; avx = fma_(q, nx, avx); avy = fma_(q, ny, avy);
                const float wsum = fma_(omb, K::r_ball, o.om * (mouth ? K::dck : K::r_robot));
                const float vt = fma_(dvy, nx, -(dvx * ny)) - wsum;
                const float lim = q * K::mu_rb;
                const float ft = clampf(vt * K::kt_rb_r, lim, -lim);
                avx = fma_(-ft, ny, avx); avy = fma_(ft, nx, avy);
                // the ball's side of the same contact
                float qb = K::ope_rb * vn * K::w_rb_b;
                const float limb = qb * K::mu_rb;
                const float ftb = clampf(vt * K::kt_rb_b, limb, -limb);
                r0.x = fma_(-ftb, ny, qb * nx); r0.y = fma_(ftb, nx, qb * ny); dws = ftb * K::spin_c; fl |= 1;
            }
            float pc = K::beta * pen * K::w_rb_r;
            apx = fma_(-pc, nx, apx); apy = fma_(-pc, ny, apy);
            float pb = K::beta * pen * K::w_rb_b; r0.z = pb * nx; r0.w = pb * ny; fl |= 2;
        }
        if (first) {
            o.ir = mouth && pen > -K::ir_tol;
            if (o.ir) {  // infrared: kicker / dribbler act on the ball
                if (o.kick_x > 0.0f || o.kick_z > 0.0f) {
                    fl |= 4 | 8;
                    r1.y = o.vx + o.kick_x * o.c; r1.z = o.vy + o.kick_x * o.s; r1.w = o.kick_z;
                } else if (o.drib) {
                    float hx = o.x + K::dck_rb * o.c, hy = o.y + K::dck_rb * o.s;
                    float cvx = (hx - ob.x) * P.drib_gain, cvy = (hy - ob.y) * P.drib_gain;
                    float m2 = cvx * cvx + cvy * cvy;
                    if (m2 > K::drib_vmax2) { float sc = K::drib_vmax / sqrtf(m2); cvx = cvx * sc; cvy = cvy * sc; }
                    fl |= 4;
                    r1.y = (o.vx - o.om * K::dck_rb * o.s) + cvx;
                    r1.z = (o.vy + o.om * K::dck_rb * o.c) + cvy;
                }
            }
        }
        r1.x = __int_as_float(fl);
        if (fl) { sh.Bq[lane] = r0; sh.Cq[lane] = r1; sh.Dq[lane] = dws; }
    }
    // which robots wrote a record: one ballot; the ball lane visits only those, in index order
    // (usually none: no LDS read at all on the ball's side)
    const unsigned long long wrote = __ballot(fl != 0);
    wave_sync();
    if (is_ball) {
        unsigned long long todo = L <= 32 ? (env_lane_mask<L>(g) & wrote) : wrote;
        while (todo) {
            const int lj = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 r1 = sh.Cq[lj];
            const float4 r0 = sh.Bq[lj];
            const int flj = __float_as_int(r1.x);
            if (flj & 1) { avx = avx - r0.x; avy = avy - r0.y; aw = aw + sh.Dq[lj]; }
            if (flj & 2) { apx = apx + r0.z; apy = apy + r0.w; got = true; }
            if (flj & 4) { bo.ovr = true; bo.okick = (flj & 8) != 0; bo.ovx = r1.y; bo.ovy = r1.z; bo.ovz = r1.w; }
        }
    }
    if (got) {   // only a body that touched something is updated (the others keep their bits)
        o.vx = o.vx + avx; o.vy = o.vy + avy;
        o.x = o.x + apx; o.y = o.y + apy;
        if (is_ball) o.om = o.om + aw;
    }
    return touched;
}

// ---------------------------------------------------------------------------------------------
// n_sub sub-steps of one env.step() for the body held by this lane.
//   b = body index of the lane (0..N-1 robots, N ball, > N idle), g = env slot in the wave
//   NR > 0: robot count known at compile time (pair loops fully unrolled); NR == 0: run-time
// ---------------------------------------------------------------------------------------------
template <int KIND, int L, int NR>
__device__ __forceinline__ void physics(const Params& P, Body& o, const int b, const int g,
                                        const bool live, Shared<L>& sh) {
    using K = KC<KIND>;
    constexpr int G = 64 / L;
    const int N = NR ? NR : P.n_robots;
    const bool is_robot = live && b < N;
    const bool is_ball = live && b == N;
    const int lane = LaneMap<L>::slot(b, g);

    // rolling resistance: a constant deceleration, applied once for the whole step() while the
    // ball is on the ground (exact stop, never reverses) — keeps the sqrt + divide chain out of
    // the sub-step loop, where the ball lane's branch is serialised with the robots' work.
    // Same place: the spin about the vertical axis decays at a constant rate to an exact stop.
    if (is_ball) ball_step_friction(P, o);
#ifdef RSX_TIMING_SUB   // development: where a sub-step's cycles go (sub-steps 1.. only; tools/exp_substep_phases.py)
    unsigned long long tsA = 0, tsB = 0, tsC = 0, ts0 = 0, ts1 = 0, ts2 = 0;
#endif

    for (int sub = 0; sub < P.n_sub; ++sub) {
#ifdef RSX_TIMING_SUB
        ts0 = __builtin_readcyclecounter();
#endif
        // ---- A: actuation + integration ----
        if (is_robot) {   // rsx_body.hpp: the per-body arithmetic is stated once for all kernel layouts
            actuate_robot<KIND>(P, o);
            o.th = advance_heading(P, o.om, o.th);
            rotate_heading(o.om * P.h, o.c, o.s);
        }
        if (RSX_RARE_B(KIND, 1, is_ball && (o.z > 0.0f || o.vz > 0.0f))) ball_flight(P, o, K::e_ground, K::vz_min);   // the ball in flight
        // the position advance is the same instruction pair for robots and the ball, outside the role branches
        // (every role branch of a lane group costs a save / branch / restore of the exec mask; idle lanes hold zeros)
        o.x = fma_(o.vx, P.h, o.x);
        o.y = fma_(o.vy, P.h, o.y);
#ifdef RSX_TIMING_SUB
        ts1 = __builtin_readcyclecounter();
#endif

        // ---- B: contacts — one Jacobi sweep over the post-integration snapshot, and a second one
        // over the corrected snapshot for the envs in which some pair overlapped by more than pen2
        // (impacts at speed, jammed piles; resting contacts stay far below).  The second sweep is
        // the same loop body again: the instructions are in the cache (a separate copy never is) ----
        // is the env's ball low enough to be touched?  One ballot of the ball lanes' answer, each lane picks its env's bit
        // (was: the height through LDS — a write, a dependent read and its wait in every sub-step)
        const unsigned long long lowm = __ballot(is_ball && o.z < K::robot_h);
        bool ball_low = ((lowm >> (LaneMap<L>::slot(N, g))) & 1ull) != 0;
        bool active = is_robot || is_ball;   // lanes whose env takes part in the current sweep
        BallOverride bo{false, false, 0.0f, 0.0f, 0.0f};
        for (int sweep = 0;; ++sweep) {
            float2 fo = float2{0.0f, 0.0f};   // VSS: this body's held axes in the snapshot of this sweep (robots; the ball publishes zeros)
            if (active) {
                sh.A[lane] = make_float4(o.x, o.y, o.vx, o.vy);
                sh.W[lane] = o.om;   // yaw rate / spin: read on the contact path only
                sh.X[g * L + b] = o.x; sh.Y[g * L + b] = o.y;
            }
            wave_sync();
            // VSS: the held axes of this snapshot are computed and published BEHIND the exchange — the arithmetic fills the wait for the
            // partners' positions (overlap_keys_packed: `fill`).  Read on the contact path only; no second exchange point is needed: a
            // wave's LDS accesses execute in issue order, and the compiler keeps this write ahead of the later reads of the same array
            // (they may alias).  (A wave_sync() at the head of the contact branch was measured: it pins the body's position in scratch
            // memory, 8.9 -> 12.8 us.)
            auto publish_held = [&]() {
                if constexpr (K::held) { fo = held_axes<KIND>(P, o.x, o.y, is_robot); sh.F[lane] = fo; }
            };
            bool deep = false;   // this lane saw a deep contact
            bool wallp = false;  // ... a touching robot - robot pair with a wall-blocked axis (model v2: wall_shares)

            if (KIND == RSX_KIND_VSS) {
                // every pair is circle-circle; only the constants depend on the pair type
                if (active) {
                    if (NR) {
                        // Overlap test of the whole sweep, exact and with ONE compare per partner class:
                        // d2 is a sum of squares (>= +0), and non-negative floats order like their bit
                        // patterns, so with u = bits(d2) - 1 (d2 == 0, the lane's own slot, wraps to
                        // 0xFFFFFFFF)   0 < d2 < thr   <=>   u < bits(thr) - 1   (unsigned).
                        // The minimum of u over the robot slots is compared once; contacts are rare, so
                        // the common case is ~5 instructions per partner and one untaken branch.
                        constexpr uint32_t T_RR = __builtin_bit_cast(uint32_t, K::rs_rr2) - 1u;
                        constexpr uint32_t T_RB = __builtin_bit_cast(uint32_t, K::rs_rb2) - 1u;
                        uint32_t u[NR + 1];
                        if constexpr (K::held) overlap_keys_packed<NR + 1, L>(sh, g, o.x, o.y, u, publish_held);
                        else overlap_keys_packed<NR + 1, L>(sh, g, o.x, o.y, u);
                        uint32_t um = u[0];
#pragma unroll
                        for (int j = 1; j < NR; ++j) um = min(um, u[j]);
                        // robot lane: robot slots are robot-robot pairs, slot NR the ball; ball lane:
                        // every robot slot is a robot-ball pair, slot NR itself (u = 0xFFFFFFFF)
                        const bool any = ((um < (is_ball ? T_RB : T_RR)) & (!is_ball | ball_low)) | ((u[NR] < T_RB) & ball_low);
                        if (RSX_RARE_B(KIND, 2, any)) {
                            // Each lane walks ITS partners in body-index order; lanes with different
                            // partners share an iteration, so a wave pays for the deepest lane (one
                            // response, rarely two) instead of one response block per distinct partner
                            // index present anywhere in the wave.  The wave that finishes last sets a
                            // single-step launch's duration, and it is always one with contacts.
                            // slot j ends at bit j: shifted in from the right, highest slot first — a compare and an add-with-
                            // carry per slot, no bit constant in a register; the ball's height gates its pairs afterwards
                            unsigned todo = 0;
                            const uint32_t thr_r = is_ball ? T_RB : T_RR;   // robot slots: robot-ball pairs for the ball lane
#pragma unroll
                            for (int j = NR; j >= 0; --j) {
                                if (j == NR) asm("v_cmp_gt_u32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(todo) : "v"(u[j]), "s"(T_RB) : "vcc");
                                else asm("v_cmp_gt_u32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(todo) : "v"(u[j]), "v"(thr_r) : "vcc");
                            }
                            if (!ball_low) todo = is_ball ? 0u : (todo & ~(1u << NR));
                            const bool v2w = K::wall_aware && __ballot(is_robot && at_wall<KIND>(P, o.x, o.y)) != 0ull;   // (rsx_body.hpp: contact_response)
                            float avx = 0.0f, avy = 0.0f, apx = 0.0f, apy = 0.0f, aw = 0.0f;
                            const float lever = is_ball ? K::r_ball : K::r_robot;
                            // software-pipelined: the next partner's slot is fetched while the current
                            // response is being computed
                            int jn = __builtin_ctz(todo);
                            todo &= todo - 1;
                            float4 nxt = sh.A[LaneMap<L>::slot(jn, g)];
                            float nxw = sh.W[LaneMap<L>::slot(jn, g)];
                            float2 nxf = sh.F[LaneMap<L>::slot(jn, g)];
                            for (;;) {
                                const int j = jn;
                                const float4 oj = nxt;
                                const float wj = nxw;
                                const float2 fj = nxf;
                                const bool more = todo != 0;
                                if (more) {
                                    jn = __builtin_ctz(todo);
                                    todo &= todo - 1;
                                    nxt = sh.A[LaneMap<L>::slot(jn, g)];
                                    nxw = sh.W[LaneMap<L>::slot(jn, g)];
                                    nxf = sh.F[LaneMap<L>::slot(jn, g)];
                                }
                                const float dx = oj.x - o.x, dy = oj.y - o.y;
                                const float d2 = fma_(dx, dx, dy * dy);   // the value the sweep above saw
                                const bool rb = is_ball || j == NR;
                                contact_response<KIND>(P, o, oj, d2, rb ? K::rs_rb : K::rs_rr, rb ? K::ope_rb : K::ope_rr,
                                                       is_ball ? K::w_rb_b : (j == NR ? K::w_rb_r : K::w_rr),
                                                       is_ball ? K::kt_rb_b : (j == NR ? K::kt_rb_r : K::kt_rr),
                                                       rb ? K::mu_rb : K::mu_rr, is_ball ? K::spin_c : 0.0f,
                                                       fma_(wj, j == NR ? K::r_ball : K::r_robot, o.om * lever), K::beta, K::pen2, !rb, v2w,
                                                       avx, avy, apx, apy, aw, deep, wallp, fo, fj);
                                if (!more) break;
                            }
                            // only a body that touched something is updated (the others keep their bits)
                            o.vx = o.vx + avx; o.vy = o.vy + avy;
                            o.x = o.x + apx; o.y = o.y + apy;
                            if (is_ball) o.om = o.om + aw;
                        }
                    } else {
                        publish_held();
                        deep = vss_sweep_loop<KIND, L>(P, o, N, g, is_ball, ball_low, sh, wallp, fo);
                    }
                }
            } else {
                deep = ssl_sweep<KIND, L, NR>(P, o, N, g, lane, is_robot && active, is_ball && active, ball_low, sweep == 0, sh, bo, wallp);
            }
            // second sweep for the envs in which some pair was deep: one ballot, usually no lane; a third and a fourth one for the
            // envs in which the last sweep also saw a wall pair (model v2: piles pressed against a wall)
            const unsigned long long dmask = __ballot(deep);
            if (sweep == 3 || !RSX_RARE_B(KIND, 2, dmask != 0)) break;
            bool again = (L == 64 ? dmask : (dmask & env_lane_mask<L>(g))) != 0;
            if (sweep >= 1) {
                const unsigned long long wmask = __ballot(wallp);
                again = again && (L == 64 ? wmask : (wmask & env_lane_mask<L>(g))) != 0;
            }
            active = active && again;
            if (sweep >= 1 && !__any(active)) break;
            wave_sync();   // every lane has read the first snapshot before it is republished
        }
        if (KIND == RSX_KIND_SSL && bo.ovr) {   // kicker / dribbler: decided in the first sweep, applied after the impulses
            o.vx = bo.ovx; o.vy = bo.ovy; o.om = 0.0f;
            if (bo.okick && bo.ovz > 0.0f) o.vz = bo.ovz;
        }

        // ---- C: walls ----
        // SSL: every lane, no role branch (idle lanes hold zeros: inside every wall) — measured 1-3 % on the SSL tasks;
        // the VSS-v0 3v3 single-step kernel measured 1.5 % slower that way and keeps the branch
        // (SSL also: only when some body of the wave is near a wall — near_walls, rsx_body.hpp: the clamp is the identity elsewhere)
#ifdef RSX_TIMING_SUB
        ts2 = __builtin_readcyclecounter();
#endif
        if (KIND == RSX_KIND_SSL ? __any(near_walls<KIND>(P, o.x, o.y)) : (is_robot || is_ball)) {
            const float vx0 = o.vx, vy0 = o.vy;
            int hit = 0;
            if constexpr (KIND == RSX_KIND_VSS) {
                // the goal-post response shares the rare branch of the ball's wall friction (one exec-mask branch at the end of every
                // sub-step instead of two: a lone wave pays for each one's compare -> scalar -> branch chain)
                const float rb = is_ball ? K::r_ball : K::r_robot, eb = is_ball ? K::e_wb : K::e_wr;
                walls<KIND, true>(P, rb, eb, o.x, o.y, o.vx, o.vy, hit);
                if (RSX_RARE_B(KIND, 2, (is_ball && (hit & 3)) || (hit & 8))) {
                    if (hit & 8) post_response(P, rb, eb, o.x, o.y, o.vx, o.vy, hit);
                    if (is_ball && (hit & 3)) ball_wall_spin<KIND>(hit, vx0, vy0, o.vx, o.vy, o.om);
                }
            } else {
            walls<KIND>(P, is_ball ? K::r_ball : K::r_robot, is_ball ? K::e_wb : K::e_wr, o.x, o.y, o.vx, o.vy, hit);
            if (RSX_RARE_B(KIND, 2, is_ball && hit)) ball_wall_spin<KIND>(hit, vx0, vy0, o.vx, o.vy, o.om);
            }
        }
        wave_sync();  // A / W / Bq / Cq / Dq are rewritten by the next sub-step
#ifdef RSX_TIMING
        if (threadIdx.x == 0 && sh.dbg) sh.dbg[(size_t)(8 + sub) * gridDim.x + blockIdx.x] = __builtin_readcyclecounter();
#endif
#ifdef RSX_TIMING_SUB
        if (sub >= 1) { const unsigned long long t3 = __builtin_readcyclecounter(); tsA += ts1 - ts0; tsB += ts2 - ts1; tsC += t3 - ts2; }
#endif
    }
#ifdef RSX_TIMING_SUB
    if (threadIdx.x == 0 && sh.dbg) {
        sh.dbg[(size_t)15 * gridDim.x + blockIdx.x] = tsA; sh.dbg[(size_t)16 * gridDim.x + blockIdx.x] = tsB; sh.dbg[(size_t)17 * gridDim.x + blockIdx.x] = tsC;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// SoA state access
// ---------------------------------------------------------------------------------------------
// Raw wire values of the lane's body.  Robot and ball lanes run the SAME load instructions (only
// the row index differs: ball rows 0..4 + the vz row, robot rows 5+RS*b .. +5), so all loads of a
// wave are in flight together; nothing is computed here (a use would park the wave on vmcnt
// between the two roles' loads).
struct RawBody { float v0, v1, v2, v3, v4, v5, ir /* ball: spin */, w[4]; };

template <int KIND>
__device__ __forceinline__ RawBody load_raw(const Params& P, const float* __restrict__ st, int e,
                                            int b, bool is_robot, bool is_ball) {
    constexpr int RS = ModelD<KIND>::rs;
    const ix_t B4 = (ix_t)4 * (ix_t)P.row_stride, e4 = (ix_t)4 * (ix_t)e;   // bytes per row, this env's column
    RawBody r{};
    if (is_robot || is_ball) {
        const int row0 = is_ball ? 0 : 5 + RS * b;
        const int row5 = is_ball ? P.state_dim : row0 + 5;
        const ix_t i0 = (ix_t)row0 * B4 + e4;
        r.v0 = at_byte(st, i0); r.v1 = at_byte(st, i0 + B4); r.v2 = at_byte(st, i0 + 2 * B4); r.v3 = at_byte(st, i0 + 3 * B4); r.v4 = at_byte(st, i0 + 4 * B4);
        r.v5 = at_byte(st, (ix_t)row5 * B4 + e4);
    }
    if (KIND == RSX_KIND_SSL) {   // robots: infrared flag; ball: spin row (same load instruction)
        if (is_robot || is_ball) r.ir = at_byte(st, (ix_t)(is_ball ? P.state_dim + 1 : 5 + RS * b + 6) * B4 + e4);
        if (is_robot) {
            const ix_t i7 = (ix_t)(5 + RS * b + 7) * B4 + e4;
#pragma unroll
            for (int i = 0; i < 4; ++i) r.w[i] = at_byte(st, i7 + (ix_t)i * B4);
        }
    } else if (is_ball) {
        r.ir = at_byte(st, (ix_t)(P.state_dim + 1) * B4 + e4);
    }
    return r;
}

// wire values -> the lane's working record (heading in degrees, rate in rad/s, exact sin / cos)
template <int KIND>
__device__ __forceinline__ void interpret_body(const RawBody& r, bool is_robot, bool is_ball, Body& o,
                                               float& th_deg, float& om_deg, float w[4]) {
    using K = KC<KIND>;
    o = Body{};
    th_deg = 0.0f; om_deg = 0.0f;
    w[0] = w[1] = w[2] = w[3] = 0.0f;
    o.x = r.v0; o.y = r.v1; o.vx = r.v3; o.vy = r.v4;
    if (is_robot) {
        th_deg = r.v2; om_deg = r.v5;
        if (KIND == RSX_KIND_SSL) {
            o.ir = r.ir != 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = r.w[i];
        }
        o.th = th_deg;
        o.om = om_deg * K::deg2rad;
        sincos_f32(o.th * K::deg2rad, o.s, o.c);
    } else if (is_ball) {
        o.z = r.v2 - K::r_ball;
        o.vz = r.v5;
        o.om = r.ir;   // spin about the vertical axis, rad/s
    }
}

// SSL wheel speeds (rad/s) implied by the body velocity — Entities/Frame.py:73-76
template <int KIND>
__device__ __forceinline__ void wheel_speeds(const Params& P, const Body& o, float w[4]) {
    using K = KC<KIND>;
    float vf = o.vx * o.c + o.vy * o.s;
    float vl = o.vy * o.c - o.vx * o.s;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = ((vl * P.wc[i] - vf * P.ws[i]) + o.om * K::r_robot) * K::inv_rw;
}

// store in wire format; th_deg / om_deg / w are the values to write for a robot.  Same row
// trick as load_raw: one store sequence for both roles.
template <int KIND>
__device__ __forceinline__ void store_body(const Params& P, float* __restrict__ st, int e, int b,
                                           bool is_robot, bool is_ball, const Body& o,
                                           float th_deg, float om_deg, const float w[4],
                                           bool write_ir) {
    using K = KC<KIND>;
    constexpr int RS = ModelD<KIND>::rs;
    const ix_t B4 = (ix_t)4 * (ix_t)P.row_stride, e4 = (ix_t)4 * (ix_t)e;
    if (is_robot || is_ball) {
        const int row0 = is_ball ? 0 : 5 + RS * b;
        const int row5 = is_ball ? P.state_dim : row0 + 5;
        const ix_t i0 = (ix_t)row0 * B4 + e4;
        at_byte(st, i0) = o.x; at_byte(st, i0 + B4) = o.y; at_byte(st, i0 + 2 * B4) = is_ball ? K::r_ball + o.z : th_deg;
        at_byte(st, i0 + 3 * B4) = o.vx; at_byte(st, i0 + 4 * B4) = o.vy;
        at_byte(st, (ix_t)row5 * B4 + e4) = is_ball ? o.vz : om_deg;
    }
    if (is_ball) at_byte(st, (ix_t)(P.state_dim + 1) * B4 + e4) = o.om;
    if (KIND == RSX_KIND_SSL && is_robot) {
        const ix_t i6 = (ix_t)(5 + RS * b + 6) * B4 + e4;
        if (write_ir) at_byte(st, i6) = o.ir ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) at_byte(st, i6 + (ix_t)(1 + i) * B4) = w[i];
    }
}

// block -> tile map: block b runs on XCD b % 8 (observed dispatch order; used for L2 affinity
// only, never for correctness), so give each XCD one contiguous range of tiles.
__device__ __forceinline__ int tile_of_block(const int per /* gridDim.x / 8: the grid is a multiple of 8 */) {
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}
// The large-batch single-step kernels walk an XCD's tiles in ALTERNATING directions from one launch to the next (the
// launcher negates `per` on odd ticks): the tiles a launch starts with are then the ones the previous launch wrote
// last, i.e. the part of the batch that still sits in the 256 MB memory-side cache (Infinity Cache survives kernel
// boundaries; the L2s do not).  Which wave steps which env changes nothing in the results.
__device__ __forceinline__ int tile_of_block_zigzag(const int per_signed) {
    const int per = per_signed < 0 ? -per_signed : per_signed;
    const int q = blockIdx.x >> 3;
    return (blockIdx.x & 7) * per + (per_signed < 0 ? per - 1 - q : q);
}
// who picks the direction: host-keyed launches get it as the sign of `per` (the launcher negates it on odd ticks);
// device-keyed launches (step_tick below) get a negative `per` for "alternate" and decide from the tick they read
__device__ __forceinline__ int zigzag_per(const bool dev, const uint32_t tick, const int per_signed) {
    if (!dev || per_signed >= 0) return per_signed;
    return (tick & 1u) ? per_signed : -per_signed;
}

// Kernel arguments.  The first twelve dwords are plain pointers / ints so that the command
// processor PRELOADS them into SGPRs (-amdgpu-kernarg-preload-count=12, gfx940+): what the first
// global loads of a wave need (base pointers, row stride, tile map) is then in registers when
// the wave starts, instead of behind a scalar-cache miss on the kernarg segment — with two waves
// per CU nearly every wave would pay that miss on its critical path.  The by-value structs that
// follow carry everything else and are fetched while the state loads are in flight.
#define RSX_HOT_ARGS float* hp_state, float* hp_aux, const float* hp_in, uint8_t* hp_flags, \
                     const int hp_num_envs, const int hp_state_dim, const int hp_per_xcd, const int hp_n_steps
// The two counts of the batch travel in the hot dwords: hp_num_envs = B, and the row pad of the [rows][B] arrays (a multiple of 64
// floats, rsx_api.hip: row_pad_for) in the upper half of hp_state_dim — scalar shifts, no wait for the parameter block.
#define RSX_HOT_DIM(state_dim, row_stride, num_envs) ((int)((unsigned)(state_dim) | ((unsigned)(((row_stride) - (num_envs)) >> 6) << 16)))
#define RSX_UNPACK_HOT(P) do { (P).num_envs = hp_num_envs; (P).state_dim = hp_state_dim & 0xFFFF; \
                               (P).row_stride = hp_num_envs + (int)(((unsigned)hp_state_dim >> 16) << 6); } while (0)
// bytes of the kernarg segment RSX_HOT_ARGS occupies: the by-value Params block starts at the next multiple of its alignment
// (the late parameter fetch below and in rsx_quad_ssl.hpp reads it from there — keep the two in step when a hot argument is added)
constexpr size_t RSX_HOT_ARGS_BYTES = 4 * sizeof(void*) + 4 * sizeof(int);
constexpr size_t RSX_PARAMS_KERNARG_OFFSET = (RSX_HOT_ARGS_BYTES + alignof(Params) - 1) / alignof(Params) * alignof(Params);
namespace hot_args_check {   // the macro and the constant cannot drift apart: a function with exactly these parameters
inline void probe(RSX_HOT_ARGS) {}
template <typename... A> constexpr size_t bytes_of(void (*)(A...)) { return (sizeof(A) + ... + 0); }
static_assert(bytes_of(&probe) == RSX_HOT_ARGS_BYTES, "RSX_HOT_ARGS changed: update RSX_HOT_ARGS_BYTES (kernarg offset of Params)");
}

// =============================================================================================
// raw simulator step: robosim.step(cmds) + get_state() on the SoA buffers
// =============================================================================================
template <int KIND, int L, int NR>
__global__ __launch_bounds__(64) void sim_step_kernel(RSX_HOT_ARGS, const Params P_, const Buffers bufs_) {
    Params P = P_; RSX_UNPACK_HOT(P);
    Buffers bufs = bufs_; bufs.state = hp_state; bufs.cmds = hp_in;   // hp_in: the command buffer
    float* const state_out = hp_aux;   // this kernel's second pointer slot: where the new state goes (== hp_state: in place)
    // fourth pointer slot: a second copy of the new state, or nullptr.  The host-format calls of small batches
    // (rsx_step / rsx_step_state: the robosim-shaped single-env path) hand in pinned host memory here and read their
    // commands from pinned host memory too: one launch + one synchronisation per step instead of copy, launch, copy
    float* const mirror = reinterpret_cast<float*>(hp_flags);
    using K = KC<KIND>;
    constexpr int G = 64 / L;
    constexpr int CD = ModelD<KIND>::cmd_dim;
    __shared__ Shared<L> sh;
#ifdef RSX_TIMING
    if (threadIdx.x == 0) sh.dbg = nullptr;
#endif
    const int lane = threadIdx.x;
    const int b = LaneMap<L>::body(lane), g = LaneMap<L>::env(lane);
    const int e = tile_of_block(hp_per_xcd) * G + g;
    const int N = NR ? NR : P.n_robots;
    const bool live = e < P.num_envs;
    const bool is_robot = live && b < N, is_ball = live && b == N;
    const size_t B = (size_t)P.num_envs;

    Body o; float od, wd, w[4];
    const RawBody raw = load_raw<KIND>(P, bufs.state, e, b, is_robot, is_ball);
    float q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int rand_tick = hp_n_steps;   // >= 0: commands are drawn here (rsx_step_dev_random), < 0: read from memory
    if (is_robot) {
        if (rand_tick >= 0) {
            const u32x4 u = philox4x32(P.env_id_base + (uint32_t)e, (uint32_t)rand_tick, (uint32_t)b, DOM_RAW, P.key0, P.key1);
            const float a0 = u01(u.x) * 2.0f - 1.0f, a1 = u01(u.y) * 2.0f - 1.0f, a2 = u01(u.z) * 2.0f - 1.0f;
            if (KIND == RSX_KIND_SSL) { q[1] = a0 * 2.5f; q[2] = a1 * 2.5f; q[3] = a2 * 10.0f; }
            else { q[0] = a0 * K::w_max; q[1] = a1 * K::w_max; }
        } else {
            const ix_t B4 = (ix_t)4 * (ix_t)P.row_stride, c0 = (ix_t)(b * CD) * B4 + (ix_t)4 * (ix_t)e;
#pragma unroll
            for (int i = 0; i < CD; ++i) q[i] = at_byte(bufs.cmds, c0 + (ix_t)i * B4);
        }
    }
    interpret_body<KIND>(raw, is_robot, is_ball, o, od, wd, w);
    if (is_robot) robot_targets<KIND>(P, o, q);
    physics<KIND, L, NR>(P, o, b, g, live, sh);
    if (is_robot) {
        od = o.th; wd = o.om * K::rad2deg;
        if (KIND == RSX_KIND_SSL) wheel_speeds<KIND>(P, o, w);
    }
    store_body<KIND>(P, state_out, e, b, is_robot, is_ball, o, od, wd, w, P.n_sub != 0 || state_out != hp_state);
    if (mirror) store_body<KIND>(P, mirror, e, b, is_robot, is_ball, o, od, wd, w, true);
}

// =============================================================================================
// fused task step
// =============================================================================================

// Episode counters (metrics[1..6]) are summed with atomics.  Device-scope atomics on ONE cache line
// serialise at about 8 ns each, whatever wave issues them: with a reset in most waves of a 10^6-env launch
// (pass endurance, contested possession) that was the whole step time (605 us instead of 82 us).  The step
// kernels therefore add into one of MSLOTS 64-byte lines, picked by block id, and fold_metrics_kernel
// (rsx_read_metrics / rsx_metrics_fold) adds the lines into metrics[] and clears them.
constexpr int MSLOTS = 256;
__device__ __forceinline__ unsigned long long* metric_slot(const Buffers& b) {
    return b.mslots + (size_t)(blockIdx.x & (MSLOTS - 1)) * RSX_METRICS;
}

// ---------------------------------------------------------------------------------------------
// The step counter that keys the per-step random draws (and the parity of the placement cache).
//
// Host-keyed (default): the host counts the stepping launches of a handle and passes the count as Params::tick_base.
// That value is baked into the launch — a hipGraph that captured the launch would replay ONE tick for ever.
// Device-keyed (rsx_task_enable_capture; flagged by RSX_TICK_DEV in the n_steps argument, a preloaded SGPR): the
// counter lives in device memory, one 32-bit slot PER WORKGROUP behind the metrics vector.  Workgroup b reads slot b
// and writes slot b + n back; launches of a handle are stream-ordered, nobody else touches that slot, so this needs no
// atomic and cannot race with late-starting workgroups of the same launch (a single shared word could: a workgroup
// that starts after another one finished would read the next launch's tick).  All slots of a handle hold the same
// value between launches (the host re-syncs the slots a smaller grid did not cover, rsx_api.hip: tick_resync).
// A counter about to wrap refuses the launch: every workgroup sees the same value, sets the error word and returns
// before touching any state (rsx.h: "checked, never wrapped").
// ---------------------------------------------------------------------------------------------
constexpr int RSX_TICK_DEV = 1 << 30;        // flag bit of the n_steps kernel argument
constexpr int RSX_N_STEPS_MASK = RSX_TICK_DEV - 1;
constexpr int TICK_ERR_WORD = 18;            // uint32 index behind metrics[0]: bytes 72..75
constexpr int TICK_SLOT_WORD0 = 64;          // uint32 index of slot 0: 256 bytes behind metrics[0]
struct StepTick { uint32_t t; bool ok; };
__device__ __forceinline__ StepTick step_tick(const bool dev, const Params& P, const Buffers& bufs, const uint32_t n) {
    if (!dev) return StepTick{P.tick_base, true};
    uint32_t* const w = reinterpret_cast<uint32_t*>(bufs.metrics);
    uint32_t* const slot = w + TICK_SLOT_WORD0 + blockIdx.x;
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)__atomic_load_n(slot, __ATOMIC_RELAXED));
    if (t > 0xFFFFFFFFu - n) {
        if (threadIdx.x == 0) w[TICK_ERR_WORD] = 1u;
        return StepTick{t, false};
    }
    if (threadIdx.x == 0) __atomic_store_n(slot, t + n, __ATOMIC_RELAXED);
    return StepTick{t, true};
}

// observation entries owned by this lane -> staging row of its env; values are the WIRE-format
// state.  Layouts: vss_gym.py:93-117, static_defenders.py:90-112, dribbling.py:76-104,
// contested_possession.py:78-104, pass_endurance.py:77-91.
// nb = blue robots (run-time for the lanes kernels, a constant for the one-lane-per-env kernels, whose
// "row" is a register array)
template <int KIND, int TASK>
__device__ __forceinline__ void write_obs_nb(const Params& P, float* __restrict__ row, int b, const int nb,
                                             bool is_robot, bool is_ball, float x, float y, float vx,
                                             float vy, float sn, float cs, float om_deg, int ir,
                                             float tscalar) {
    // sn / cs = sin / cos of (theta_deg * deg2rad), i.e. of the wire-format heading
    using T = TC<TASK>;
    const float lo = -1.2f, hi = 1.2f;
    if (TASK == RSX_TASK_SSL_SCRIMMAGE) {   // positions only (README.md:88-90 style)
        if (is_ball || is_robot) {
            float* r = row + (is_ball ? 0 : 2 + 2 * b);
            r[0] = clampf(x * P.inv_max_pos, lo, hi);
            r[1] = clampf(y * P.inv_max_pos, lo, hi);
        }
        return;
    }
    constexpr int OFF = TASK == RSX_TASK_SSL_DRIBBLING ? 1 : 0;   // dribbling: slot 0 = checkpoint progress
    constexpr int WB = TASK == RSX_TASK_VSS_V0 ? 7 : (TASK == RSX_TASK_SSL_PASS_ENDURANCE ? 6 : 8);
    constexpr int WY = TASK == RSX_TASK_VSS_V0 ? 5 : 2;
    if (is_ball) {
        if (OFF) row[0] = ((tscalar / 6.0f) * 2.0f) - 1.0f;
        row[OFF + 0] = clampf(x * P.inv_max_pos, lo, hi);
        row[OFF + 1] = clampf(y * P.inv_max_pos, lo, hi);
        row[OFF + 2] = clampf(vx * T::inv_max_v, lo, hi);
        row[OFF + 3] = clampf(vy * T::inv_max_v, lo, hi);
    } else if (is_robot) {
        if (b < nb) {
            float* r = row + OFF + 4 + WB * b;
            r[0] = clampf(x * P.inv_max_pos, lo, hi);
            r[1] = clampf(y * P.inv_max_pos, lo, hi);
            r[2] = sn; r[3] = cs;
            if (TASK == RSX_TASK_SSL_PASS_ENDURANCE) {
                r[4] = clampf(om_deg * T::inv_max_w, lo, hi);
                r[5] = ir ? 1.0f : 0.0f;
            } else {
                r[4] = clampf(vx * T::inv_max_v, lo, hi);
                r[5] = clampf(vy * T::inv_max_v, lo, hi);
                r[6] = clampf(om_deg * T::inv_max_w, lo, hi);
                if (TASK == RSX_TASK_SSL_DRIBBLING) r[7] = ir ? 1.0f : -1.0f;
                else if (TASK != RSX_TASK_VSS_V0) r[7] = ir ? 1.0f : 0.0f;
            }
        } else {
            float* r = row + OFF + 4 + WB * nb + WY * (b - nb);
            r[0] = clampf(x * P.inv_max_pos, lo, hi);
            r[1] = clampf(y * P.inv_max_pos, lo, hi);
            if (TASK == RSX_TASK_VSS_V0) {
                r[2] = clampf(vx * T::inv_max_v, lo, hi);
                r[3] = clampf(vy * T::inv_max_v, lo, hi);
                r[4] = clampf(om_deg * T::inv_max_w, lo, hi);
            }
        }
    }
}

template <int KIND, int TASK>
__device__ __forceinline__ void write_obs(const Params& P, float* __restrict__ row, int b,
                                          bool is_robot, bool is_ball, float x, float y, float vx,
                                          float vy, float sn, float cs, float om_deg, int ir,
                                          float tscalar) {
    write_obs_nb<KIND, TASK>(P, row, b, P.n_blue, is_robot, is_ball, x, y, vx, vy, sn, cs, om_deg, ir, tscalar);
}

// Return of a finished VSS-v0 episode, from its cumulative reward terms (vss_gym.py:151-158,186-190):
// shaping sums + 10 per goal for, -10 per goal against.  (No running sum of rewards is kept.)
__device__ __forceinline__ float vss_episode_return(const float* info) {
    return ((info[1] + info[2]) + info[3]) + 10.0f * info[0];
}

// vss_gym.py:235-254
__device__ __forceinline__ float vss_wheel(float a) {
    using T = TC<RSX_TASK_VSS_V0>;
    using K = KC<RSX_KIND_VSS>;
    float v = a * T::max_v;
    v = clampf(v, -T::max_v, T::max_v);
    if (-T::deadzone < v && v < T::deadzone) v = 0.0f;
    return v * K::inv_rw;
}

// Action of the agent (blue 0) of an SSL task -> its robot command q (robosim order: wheel speeds flag,
// v_x, v_y, v_theta, kick_x, kick_z is q[5]..., dribbler q[7]); (sn, cs) = sine and cosine of the robot's heading: the
// body's own (s, c), which every step start and end derive from the stored heading in degrees by sincos_f32 — the
// reference evaluates sin / cos of that same float (static_defenders.py:128-131).
template <int TASK>
__device__ __forceinline__ void ssl_agent_commands(const float* a, const float sn, const float cs, float* q) {
    using K = KC<RSX_KIND_SSL>;
    using T = TC<TASK>;
    if (TASK == RSX_TASK_SSL_PASS_ENDURANCE) {  // pass_endurance.py:106-130
        float k = fabsf(a[1]) > 0.5f ? a[1] : 0.0f;
        q[3] = a[0] * 10.0f;
        q[5] = k * 5.0f;
        q[7] = a[2] > 0.0f ? 1.0f : 0.0f;
    } else {  // static_defenders.py:114-148, dribbling.py:106-135, contested_possession.py:106-137
        float gx = a[0] * T::max_v, gy = a[1] * T::max_v, vth = a[2] * 10.0f;
        float lx = gx * cs + gy * sn, ly = gy * cs - gx * sn;
        float nrm = sqrtf(lx * lx + ly * ly);
        if (!(nrm < T::max_v)) { float sc = T::max_v / nrm; lx = lx * sc; ly = ly * sc; }
        q[1] = lx; q[2] = ly; q[3] = vth;
        if (TASK == RSX_TASK_SSL_DRIBBLING) {
            q[7] = a[3] > 0.0f ? 1.0f : 0.0f;
        } else {
            q[5] = a[3] > 0.0f ? 5.0f : 0.0f;
            q[7] = a[4] > 0.0f ? 1.0f : 0.0f;
        }
    }
}

// Reward, termination and info terms of one env step, from the post-step ball position (bx, by), the
// pre-step one (lastx, lasty) and xr[] = what the task needs from the robots (filled by the caller: see
// task_step_kernel).  One body for both tile layouts, so their arithmetic cannot drift apart.
template <int KIND, int TASK>
__device__ __forceinline__ void task_reward(const Params& P, const float* xr, const float bx, const float by,
                                            const float lastx, const float lasty, const bool first_step,
                                            float& prev_pot, float* info, float& reward, int& term,
                                            bool& success, bool& against) {
    using T = TC<TASK>;
    reward = 0.0f; term = 0;
    if (TASK == RSX_TASK_VSS_V0) {  // vss_gym.py:144-192,256-311
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
            float rbx = bx - xr[0], rby = by - xr[1];
            float nrm = sqrtf(rbx * rbx + rby * rby);
            float mv = nrm > 0.0f ? (rbx / nrm) * xr[2] + (rby / nrm) * xr[3] : 0.0f;   // unguarded in vss_gym.py:298
            float move = clampf(mv * 2.5f, -5.0f, 5.0f);
            float energy = -(fabsf(xr[4]) + fabsf(xr[5]));
            float t_move = 0.2f * move, t_grad = 0.8f * grad, t_en = 2e-4f * energy;
            reward = (t_move + t_grad) + t_en;
            info[1] += t_move; info[2] += t_grad; info[3] += t_en;
        }
    } else if (TASK == RSX_TASK_SSL_SCRIMMAGE) {  // README.md:96-102 style: a goal ends the episode
        if (bx > P.half_len && fabsf(by) < P.ghw) { reward = 1.0f; term = 1; info[0] += 1.0f; }
        else if (bx < -P.half_len && fabsf(by) < P.ghw) { reward = -1.0f; term = 1; info[1] += 1.0f; }
        success = info[0] > 0.0f; against = info[1] > 0.0f;
    } else if (TASK == RSX_TASK_SSL_DRIBBLING) {  // dribbling.py:137-185; prev_pot = checkpoints_count
        const float rx = xr[0], ry = xr[1];
        if (xr[2] != 0.0f || xr[3] != 0.0f || xr[4] != 0.0f || xr[5] != 0.0f) term = 1;  // an obstacle was hit
        if (rx < -3.0f || rx > 1.0f || fabsf(ry) > 1.0f) term = 1;                         // left the course
        else {
            const int n = (int)prev_pot;
            const bool down = lasty >= 0.0f && by < 0.0f, up = lasty < 0.0f && by >= 0.0f;
            bool passed;
            if (n == 0) passed = bx < -0.5f && bx > -1.0f && down;
            else if (n == 1) passed = bx < -1.0f && bx > -1.5f && up;
            else if (n % 2 == 0) {
                const bool inside = bx < -1.5f && bx > -2.0f;
                passed = inside && down;
                if (inside && !down && up) term = 1;   // reversed the last checkpoint
            } else passed = bx > -3.0f && bx < -2.0f && up;
            if (passed) {
                reward = 1.0f;
                prev_pot = (float)(n + 1);
                if (n >= 2 && n % 2 == 0 && n + 1 == 7) term = 1;   // course completed
            }
        }
        info[0] = prev_pot;
        success = prev_pot >= 7.0f;
    } else if (TASK == RSX_TASK_SSL_PASS_ENDURANCE) {  // pass_endurance.py:132-154,187-233; prev_pot = stopped_steps
        const float shx = xr[0], shy = xr[1], rcx = xr[2], rcy = xr[3];
        const bool rc_ir = xr[4] != 0.0f;
        float ddx = rcx - bx, ddy = rcy - by, ldx = rcx - lastx, ldy = rcy - lasty;
        float dist = sqrtf(ddx * ddx + ddy * ddy), last_dist = sqrtf(ldx * ldx + ldy * ldy);
        if (rc_ir) { reward = 1.0f; term = 1; }
        else {
            float gr = P.inv_bg_scale * clampf(last_dist - dist, -1.0f, 1.0f);
            reward = gr; info[1] += gr;
        }
        // "wrong ball": outside the shooter-receiver box on a centimetre grid, or stalled
        const int cbx = (int)(bx * 100.0f), cby = (int)(by * 100.0f);
        const int csx = (int)(shx * 100.0f), csy = (int)(shy * 100.0f);
        const int crx = (int)(rcx * 100.0f), cry = (int)(rcy * 100.0f);
        const bool in_x = min(crx, csx) <= cbx && cbx <= max(crx, csx);
        const bool in_y = min(cry, csy) <= cby && cby <= max(cry, csy);
        if (fabsf(last_dist - dist) < 0.01f) prev_pot = prev_pot + 1.0f; else prev_pot = 0.0f;
        if (prev_pot > 20.0f || !(in_x && in_y)) { reward = reward - 1.0f; term = 1; }
        if (term) {
            float rdx = rcx - shx, rdy = rcy - shy;
            float dist_robs = sqrtf(rdx * rdx + rdy * rdy);
            info[0] = dist_robs > 0.0f ? (dist_robs - dist) / dist_robs : 0.0f;
        }
        success = term && rc_ir;
    } else {  // static_defenders.py:150-212,256-322; contested_possession.py:139-201
        const float rx = xr[0], ry = xr[1];
        if (TASK == RSX_TASK_SSL_CONTESTED && xr[2] != 0.0f) { info[8] += 1.0f; term = 1; }  // opponent moved
        if (rx < -0.2f || fabsf(ry) > P.half_wid) { term = 1; info[4] += 1.0f; }
        else if (rx > P.pen_x && fabsf(ry) < P.half_pen_wid) { term = 1; info[1] += 1.0f; }
        else if (bx < 0.0f || fabsf(by) > P.half_wid) { term = 1; info[2] += 1.0f; }
        else if (bx > P.half_len) {
            term = 1;
            if (fabsf(by) < P.ghw) { reward = 5.0f; info[0] += 1.0f; }
            else info[3] += 1.0f;
        } else {
            float ldx = xr[6] - lastx, ldy = xr[7] - lasty;
            float cdx = rx - bx, cdy = ry - by;
            float bd = clampf(sqrtf(ldx * ldx + ldy * ldy) - sqrtf(cdx * cdx + cdy * cdy), -1.0f, 1.0f) * P.inv_bd_scale;
            float lgx = P.half_len - lastx, cgx = P.half_len - bx;
            float bg = clampf(sqrtf(lgx * lgx + lasty * lasty) - sqrtf(cgx * cgx + by * by), -1.0f, 1.0f) * P.inv_bg_scale;
            float en = -(((fabsf(xr[8]) + fabsf(xr[9])) + fabsf(xr[10])) + fabsf(xr[11])) * T::inv_en_scale;
            info[5] += bd; info[6] += bg; info[7] += en;
            reward = (bd + bg) + en;
        }
        success = info[0] > 0.0f;
    }
}

// Random placement of one env (vss_gym.py:194-233 / static_defenders.py:214-254 with Philox
// draws).  The algorithm is sequential (every candidate is tested against the bodies already
// placed, and the index of a draw depends on how many were rejected before it), and runs on the
// env's ball lane; the expensive part, the Philox blocks, is hoisted: draws 0..NPRE-1 were
// computed speculatively by all lanes of the env (place_predraw) and are only read here.
// Poses go to A[body slot] = (x, y, theta_deg, 0).
// how many placement draws are computed ahead by the env's lanes: the rejection-sampled tasks
// use >= 13 (VSS-v0) / >= 15 (static defenders), pass endurance two plus its rejections (about
// half of its candidates; measured faster with the full block than with 4 or 8); contested
// possession uses exactly one, dribbling none (fixed course).  Later draws are computed where
// they are needed.
template <int TASK, int L>
__host__ __device__ constexpr int predraw_count() {
    return (TASK == RSX_TASK_SSL_DRIBBLING || TASK == RSX_TASK_SSL_SCRIMMAGE) ? 0 : TASK == RSX_TASK_SSL_CONTESTED ? 1 : (L < 16 ? 16 : L);
}

template <int TASK, int L>
__device__ __forceinline__ void place_predraw(const Params& P, uint32_t env_id, uint32_t episode,
                                              int b, float2* __restrict__ draws) {
    constexpr int NPRE = predraw_count<TASK, L>();
#pragma unroll
    for (int n = b; n < NPRE; n += L) {
        const u32x4 u = philox4x32(env_id, episode, (uint32_t)n, DOM_PLACE, P.key0, P.key1);
        draws[n] = make_float2(u01(u.x), u01(u.y));
    }
}

template <int TASK, int L, bool PRE = true>
__device__ __forceinline__ void place_env(const Params& P, const int N, uint32_t env_id,
                                          uint32_t episode, int g, float4* A, const float2* draws) {
    constexpr int G = 64 / L;
    constexpr int NPRE = PRE ? predraw_count<TASK, L>() : 0;   // !PRE: every draw is computed where it is used
    uint32_t n = 0;
    auto draw = [&]() -> float2 {
        const uint32_t i = n++;
        if (i < (uint32_t)NPRE) return draws[i];
        const u32x4 u = philox4x32(env_id, episode, i, DOM_PLACE, P.key0, P.key1);
        return make_float2(u01(u.x), u01(u.y));
    };
    int first = 0;
    float bx, by;
    if (TASK == RSX_TASK_SSL_DRIBBLING) {  // dribbling.py:187-202: fixed course
        A[LaneMap<L>::slot(N, g)] = make_float4(-0.1f, 0.0f, 0.0f, 0.0f);
        A[LaneMap<L>::slot(0, g)] = make_float4(0.0f, 0.0f, 180.0f, 0.0f);
        for (int k = 1; k < 5; ++k) A[LaneMap<L>::slot(k, g)] = make_float4(-0.5f * (float)k, 0.0f, 180.0f, 0.0f);
        return;
    }
    if (TASK == RSX_TASK_SSL_CONTESTED) {  // contested_possession.py:203-220: the opponent holds the ball
        const float2 u = draw();
        const float ex = P.pl_xlo + P.pl_xspan * u.x, ey = P.pl_ylo + P.pl_yspan * u.y;
        A[LaneMap<L>::slot(N, g)] = make_float4(ex - 0.1f, ey, 0.0f, 0.0f);
        A[LaneMap<L>::slot(0, g)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        A[LaneMap<L>::slot(1, g)] = make_float4(ex, ey, 180.0f, 0.0f);
        return;
    }
    if (TASK == RSX_TASK_SSL_PASS_ENDURANCE) {  // pass_endurance.py:156-185
        const float2 u = draw();
        const float px = -1.5f + 3.0f * u.x, py = 1.5f + -3.0f * u.y;
        const float side = py < 0.0f ? -1.0f : 1.0f;
        const float sx = px, sy = py + 0.115f * side;
        float rx = 0.0f;
        for (int t = 0; t < 64; ++t) {
            const float2 v = draw();
            rx = -1.5f + 3.0f * v.x;
            if (!(fabsf(rx - px) < 1.0f)) break;
        }
        const float ry = -py;
        A[LaneMap<L>::slot(N, g)] = make_float4(px, py, 0.0f, 0.0f);
        A[LaneMap<L>::slot(0, g)] = make_float4(sx, sy, side > 0.0f ? 270.0f : 90.0f, 0.0f);
        A[LaneMap<L>::slot(1, g)] = make_float4(rx, ry, (atan2_f32(ry - sy, rx - sx) + 3.14159265358979323846f) * KC<RSX_KIND_SSL>::rad2deg, 0.0f);
        return;
    }
    if (TASK == RSX_TASK_SSL_STATIC_DEFENDERS) {
        bx = 0.0f; by = 0.0f;
        for (int t = 0; t < 64; ++t) {
            const float2 u = draw();
            bx = P.pl_xlo + P.pl_xspan * u.x;
            by = P.pl_ylo + P.pl_yspan * u.y;
            if (!(bx > P.pen_x && fabsf(by) < P.half_pen_wid)) break;
        }
        A[LaneMap<L>::slot(0, g)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // blue 0 at the origin
        first = 1;
    } else {
        const float2 u = draw();
        bx = P.pl_xlo + P.pl_xspan * u.x;
        by = P.pl_ylo + P.pl_yspan * u.y;
    }
    A[LaneMap<L>::slot(N, g)] = make_float4(bx, by, 0.0f, 0.0f);
    for (int k = first; k < N; ++k) {
        float x = 0.0f, y = 0.0f;
        for (int t = 0; t < 64; ++t) {
            const float2 u = draw();
            x = P.pl_xlo + P.pl_xspan * u.x;
            y = P.pl_ylo + P.pl_yspan * u.y;
            bool ok = true;
            {   // ball first, then (static defenders) blue 0, then the robots placed so far
                float dx = x - bx, dy = y - by;
                if (dx * dx + dy * dy < P.pl_min_d2) ok = false;
            }
            for (int q = 0; q < k; ++q) {
                const float4 pq = A[LaneMap<L>::slot(q, g)];
                float dx = x - pq.x, dy = y - pq.y;
                if (dx * dx + dy * dy < P.pl_min_d2) ok = false;
            }
            if (ok) break;
        }
        const float2 u = draw();
        A[LaneMap<L>::slot(k, g)] = make_float4(x, y, 360.0f * u.x, 0.0f);
    }
}

// The same placement, run by all lanes of the env together (VSS-v0 and static defenders, whose
// placements are rejection loops).  The sequential algorithm consumes draws in order, so the
// draw index of robot k depends on how many candidates were rejected before it; here every
// robot k >= m (m = first robot not yet fixed) proposes its candidate ASSUMING no further
// rejection (draw n + 2(k-m) for the position, the next one for theta), each tests itself against
// the ball and all robots q < k (fixed or proposed), and the lowest failing robot f decides:
// m..f-1 were tested against accepted poses only, exactly as the sequential loop would have, and
// are fixed; f has used one more try and the draw indices behind it shift by one; robots > f
// propose again.  One round per rejection (+1) instead of ~2N dependent LDS round trips on a
// single lane; draws, candidates, tests and therefore results are those of place_env.
template <int TASK, int L, int NRC>
__device__ __forceinline__ float4 place_env_parallel(const Params& P, const int N, const uint32_t env_id,
                                                      const uint32_t episode, const int b, const int g,
                                                      const bool is_robot, float4* A, const float2* draws) {
    constexpr int G = 64 / L;
    constexpr int NPRE = predraw_count<TASK, L>();
    auto getdraw = [&](uint32_t i) -> float2 {
        if (i < (uint32_t)NPRE) return draws[i];
        const u32x4 u = philox4x32(env_id, episode, i, DOM_PLACE, P.key0, P.key1);
        return make_float2(u01(u.x), u01(u.y));
    };
    uint32_t n = 0;
    float bx = 0.0f, by = 0.0f;
    int m = 0;
    float x = 0.0f, y = 0.0f, th = 0.0f;   // this lane's robot
    if (TASK == RSX_TASK_SSL_STATIC_DEFENDERS) {
        for (int t = 0; t < 64; ++t) {
            const float2 u = getdraw(n++);
            bx = P.pl_xlo + P.pl_xspan * u.x;
            by = P.pl_ylo + P.pl_yspan * u.y;
            if (!(bx > P.pen_x && fabsf(by) < P.half_pen_wid)) break;
        }
        if (b == 0) A[LaneMap<L>::slot(0, g)] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // blue 0 at the origin
        m = 1;
    } else {
        const float2 u = getdraw(n++);
        bx = P.pl_xlo + P.pl_xspan * u.x;
        by = P.pl_ylo + P.pl_yspan * u.y;
    }
    // lanes of this env: body j sits at lane LaneMap<L>::slot(j, g)
    const unsigned long long envmask = env_lane_mask<L>(g);
    int t = 0;   // tries robot m has used
    while (m < N) {
        const bool spec = is_robot && b >= m;
        if (spec) {
            const uint32_t c = n + 2u * (uint32_t)(b - m);
            const float2 u = getdraw(c), v = getdraw(c + 1u);
            x = P.pl_xlo + P.pl_xspan * u.x;
            y = P.pl_ylo + P.pl_yspan * u.y;
            th = 360.0f * v.x;
            A[LaneMap<L>::slot(b, g)] = make_float4(x, y, th, 0.0f);
        }
        wave_sync();
        bool bad = false;
        if (spec) {
            {
                float dx = x - bx, dy = y - by;
                if (dx * dx + dy * dy < P.pl_min_d2) bad = true;
            }
            if (NRC) {
                float4 pq[NRC ? NRC : 1];
#pragma unroll
                for (int q = 0; q < NRC; ++q) pq[q] = A[LaneMap<L>::slot(q, g)];
#pragma unroll
                for (int q = 0; q < NRC; ++q) {
                    float dx = x - pq[q].x, dy = y - pq[q].y;
                    if ((q < b) & (dx * dx + dy * dy < P.pl_min_d2)) bad = true;
                }
            } else {
                for (int q = 0; q < b; ++q) {
                    const float4 pq = A[LaneMap<L>::slot(q, g)];
                    float dx = x - pq.x, dy = y - pq.y;
                    if (dx * dx + dy * dy < P.pl_min_d2) bad = true;
                }
            }
            if (b == m && t == 63) bad = false;   // the 64th candidate is taken as it is
        }
        wave_sync();   // the next round overwrites A
        const unsigned long long bm = __ballot(bad) & envmask;
        const int f = bm ? LaneMap<L>::body((int)__builtin_ctzll(bm)) : N;   // lowest failing robot
        if (f < N) {
            t = f == m ? t + 1 : 1;
            n += 2u * (uint32_t)(f - m) + 1u;
        }
        m = f;
    }
    return is_robot ? make_float4(x, y, th, 0.0f) : make_float4(bx, by, 0.0f, 0.0f);
}

// The random numbers of one step for the body of this lane.  They depend on (seed, global env id,
// handle step count) only — not on anything in memory — so a single-step launch computes them while
// its state loads are in flight (Philox + Box-Muller: ~1.5 k cycles that used to follow the ~1.8 k
// cycle load wait).  VSS-v0: robot 0 -> two uniforms in [-1, 1) (its random action), robots >= 1 ->
// two standard normals (Box-Muller, Utils/Utils.py:18); scrimmage: four uniforms per robot; the other
// SSL tasks: up to five uniforms for robot 0.
struct StepDraw { float v[5]; };

template <int KIND, int TASK>
__device__ __forceinline__ StepDraw draw_for_step(const Params& P, const uint32_t env_id, const uint32_t t,
                                                  const int b, const bool is_robot, const bool fed) {
    StepDraw d;
#pragma unroll
    for (int i = 0; i < 5; ++i) d.v[i] = 0.0f;
    if (TASK == RSX_TASK_VSS_V0) {
        if (is_robot && !(fed && b == 0)) {
            // one Philox call per lane: block b >> 1 of the step, this robot's pair of words
            const u32x4 u = philox4x32(env_id, 0u, t, DOM_ACT | ((uint32_t)(b >> 1) << 8), P.key0, P.key1);
            const uint32_t w0 = (b & 1) ? u.z : u.x, w1 = (b & 1) ? u.w : u.y;
            if (b == 0) { d.v[0] = u01(w0) * 2.0f - 1.0f; d.v[1] = u01(w1) * 2.0f - 1.0f; }
            else {
                float u1 = (float)((w0 >> 8) + 1u) * 5.9604644775390625e-08f;
                float ang = (u01(w1) - 0.5f) * 6.283185307179586f;
                float rad = sqrtf(-2.0f * log_f32(u1));
                float sn, cs;
                sincos_f32(ang, sn, cs);
                d.v[0] = rad * cs; d.v[1] = rad * sn;
            }
        }
    } else if (TASK == RSX_TASK_SSL_SCRIMMAGE) {
        if (is_robot && !fed) {
            const u32x4 u = philox4x32(env_id, 0u, t, DOM_ACT | ((uint32_t)b << 8), P.key0, P.key1);
            d.v[0] = u01(u.x) * 2.0f - 1.0f; d.v[1] = u01(u.y) * 2.0f - 1.0f;
            d.v[2] = u01(u.z) * 2.0f - 1.0f; d.v[3] = u01(u.w) * 2.0f - 1.0f;
        }
    } else {
        if (is_robot && b == 0 && !fed) {
            const u32x4 u = philox4x32(env_id, 0u, t, DOM_ACT, P.key0, P.key1);
            d.v[0] = u01(u.x) * 2.0f - 1.0f; d.v[1] = u01(u.y) * 2.0f - 1.0f;
            d.v[2] = u01(u.z) * 2.0f - 1.0f; d.v[3] = u01(u.w) * 2.0f - 1.0f;
            // fifth component: the low bytes u01 leaves unused in x, y, z (one block per step)
            const uint32_t w = (u.x & 0xFFu) | ((u.y & 0xFFu) << 8) | ((u.z & 0xFFu) << 16);
            d.v[4] = u01(w << 8) * 2.0f - 1.0f;
        }
    }
    return d;
}

// ---------------------------------------------------------------------------------------------
// Placement cache (single-step launches of SSLStaticDefenders 1v6 at latency-bound batch sizes).
//
// A single-step launch lasts as long as its slowest wave, and with short episodes that wave is one that resets an
// env: Philox blocks + rejection rounds are ~2.7 k cycles on top of a ~17 k cycle wave (1v6 at 2048 envs: 5.5 waves
// per launch hold a reset).  But the placement of an env's NEXT episode is a pure function of (seed, global env id,
// episode + 1): it can be computed at any time before it is needed, by anybody.  At these batches half of the chip's
// SIMDs are idle, so every step launch carries ceil(B / 64) extra HELPER workgroups behind the tile workgroups: helper
// w looks at envs [64 w, 64 w + 64) and, where the cached pose set is not the one of episode + 1, computes it with the
// same code the reset path runs (place_predraw + place_env_parallel: same draws, same tests, same poses) — eight envs
// at a time, far shorter than a step, never the slowest wave.  The resetting wave then only copies three floats per
// body that it loaded with its state.
//
// Two buffers, alternating by step parity: launch t writes buffer (t & 1) and reads buffer ((t + 1) & 1), which
// nobody writes during launch t — the only synchronisation is the kernel boundary.  An entry is tagged with the episode
// id it was made for; a tag that does not match (first steps after a reset, two episode ends in consecutive steps, a
// restored checkpoint) sends the env down the inline path, which stays as it was.  Layouts that do not use the cache
// ignore it: it is derived data, not state (not part of a checkpoint).
// Buffer: rows c * (N + 1) + b for c = x, y, theta and body b (ball = N), then the tag row; [rows][B] floats.
// ---------------------------------------------------------------------------------------------
template <int NB>
__host__ __device__ constexpr int pcache_rows() { return 3 * NB + 1; }

template <int KIND, int L, int TASK, int NR>
__device__ __forceinline__ void placement_helper(const Params& P, const Buffers& bufs, const int helper, const uint32_t tick, Shared<L>& sh) {
    static_assert(L == 8 && NR > 0, "the placement cache serves the 8-lane kernels of the fixed team sizes");
    constexpr int G = 64 / L, N = NR, NBD = N + 1;
    const size_t B = (size_t)P.num_envs;
    const int lane = threadIdx.x;
    const int b = LaneMap<L>::body(lane), g = LaneMap<L>::env(lane);
    float* const pw = bufs.pcache + (size_t)(tick & 1u) * (size_t)pcache_rows<NBD>() * B;
    // one lane per env: which of this wave's 64 envs lack the poses of their next episode?
    const int e0 = helper * 64 + lane;
    uint32_t ep_next = 0;
    bool stale = false;
    if (e0 < P.num_envs) {
        ep_next = __float_as_uint(bufs.aux[(size_t)ROW_EPISODE * (size_t)P.row_stride + e0]) + 1u;
        stale = __float_as_uint(pw[(size_t)(3 * NBD) * B + e0]) != ep_next;
    }
    unsigned long long todo = __ballot(stale);
    if (todo == 0) return;
    sh.ep[lane] = ep_next;
    wave_sync();
    while (todo) {   // eight envs per round: env slot g takes the g-th stale env
        unsigned long long mine = todo;
        for (int i = 0; i < g; ++i) mine &= mine - 1;
        const bool has = mine != 0;
        const int idx = has ? (int)__builtin_ctzll(mine) : 0;
        for (int i = 0; i < G && todo; ++i) todo &= todo - 1;
        const int e = helper * 64 + idx;
        const uint32_t env_id = P.env_id_base + (uint32_t)e;
        const uint32_t episode = sh.ep[idx];
        const bool is_robot = has && b < N, is_ball = has && b == N;
        if (has) place_predraw<TASK, L>(P, env_id, episode, b, sh.draws[g]);
        wave_sync();
        float4 pz = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (has) pz = place_env_parallel<TASK, L, NR>(P, N, env_id, episode, b, g, is_robot, sh.A, sh.draws[g]);
        if (is_robot || is_ball) {
            pw[(size_t)(0 * NBD + b) * B + e] = pz.x; pw[(size_t)(1 * NBD + b) * B + e] = pz.y; pw[(size_t)(2 * NBD + b) * B + e] = pz.z;
        }
        if (is_ball) pw[(size_t)(3 * NBD) * B + e] = __uint_as_float(episode);
        wave_sync();   // draws / A are rewritten by the next round
    }
}

// MODE (compile-time, so the per-step launch carries no loop and none of the reset-only code):
//   MODE_STEP    one step(action) per launch
//   MODE_ROLLOUT n_steps random-action steps per launch (state stays in registers)
//   MODE_RESET   reset() with random placement
//   MODE_REFRESH open a new episode on the state already in the buffers for the envs flagged in
//                the third row of the flags array (reset_to); their observations recomputed, state untouched
constexpr int MODE_STEP = 0, MODE_RESET = 1, MODE_REFRESH = 2, MODE_ROLLOUT = 3;

#ifndef RSX_TASK_KERNEL_ATTR
#define RSX_TASK_KERNEL_ATTR   // (rsx_big.hip sets an occupancy target for its build of this kernel)
#endif
template <int KIND, int L, int TASK, int NR, int MODE>
__global__ __launch_bounds__(64) RSX_TASK_KERNEL_ATTR void task_step_kernel(RSX_HOT_ARGS, const Params P_, const Buffers bufs_) {
    // A multi-step launch is short of SGPRs, not of start-up latency: there the preloaded copies
    // are left dead and everything is fetched from the kernarg segment when it is needed.
    constexpr bool HOT = MODE != MODE_ROLLOUT;
    Params P = P_;
    Buffers bufs = bufs_;
    if (HOT) {
        RSX_UNPACK_HOT(P);
        bufs.state = hp_state; bufs.aux = hp_aux; bufs.actions = hp_in; bufs.flags = hp_flags;
    }
    const int n_steps_arg = hp_n_steps & RSX_N_STEPS_MASK;
    constexpr int mode = MODE == MODE_ROLLOUT ? MODE_STEP : MODE;
    const int n_steps = MODE == MODE_ROLLOUT ? n_steps_arg : 1;
    // the step counter of this launch (see step_tick): every workgroup — tiles, idle tail, placement helpers — takes part
    const bool tick_dev = __builtin_expect((hp_n_steps & RSX_TICK_DEV) != 0, 0);
    using K = KC<KIND>;
    using T = TC<TASK>;
    constexpr int G = 64 / L;
    constexpr int ID = T::info_dim;
    constexpr int AD = T::act_dim;
    __shared__ Shared<L> sh;
    // placement cache: single-step launches of the two rejection-sampled tasks in their fixed-size 8-lane variants
    // (VSS-v0 3v3 would qualify as well and was measured: five resets per 4096-env launch — its slowest wave is a contact wave,
    // 9.33 vs 9.30-9.38 us — for 112 B more reads per env-step; it keeps the inline placement)
    constexpr bool PC = MODE == MODE_STEP && L == 8 && TASK == RSX_TASK_SSL_STATIC_DEFENDERS && NR == 7;
    if constexpr (PC) {
        if (__builtin_expect(bufs.pcache != nullptr && (int)blockIdx.x >= hp_per_xcd * 8, 0)) {   // a helper workgroup (behind the tiles)
            const StepTick th = step_tick(tick_dev, P, bufs, 1u);
            if (th.ok) placement_helper<KIND, L, TASK, (PC ? NR : 1)>(P, bufs, (int)blockIdx.x - hp_per_xcd * 8, th.t, sh);
            return;
        }
    }
    StepTick tk{0u, true};
    if (MODE == MODE_STEP || MODE == MODE_ROLLOUT) {
        tk = step_tick(tick_dev, P, bufs, (uint32_t)n_steps);
        if (__builtin_expect(!tk.ok, 0)) return;
    }
    const uint32_t tick0 = tk.t;
    const int lane = threadIdx.x;
    const int b = LaneMap<L>::body(lane), g = LaneMap<L>::env(lane);
    const int tile = tile_of_block(HOT ? hp_per_xcd : (int)(gridDim.x >> 3));
    const int e = tile * G + g;
    const int N = NR ? NR : P.n_robots;
    const bool live = e < P.num_envs;
    const bool is_robot = live && b < N, is_ball = live && b == N;
    const size_t B = (size_t)P.num_envs;
    const uint32_t env_id = P.env_id_base + (uint32_t)e;
    // observation width: a compile-time constant when the team sizes are (lets the copy-out unroll)
    constexpr int OD_C = NR == 0 ? 0
        : TASK == RSX_TASK_VSS_V0 ? 4 + 6 * NR                // equal teams: 4 + 7*nb + 5*ny (vss_gym.py:64-67): 40 for 3v3, 64 for 5v5
        : TASK == RSX_TASK_SSL_STATIC_DEFENDERS ? 4 + 8 + 2 * (NR - 1)
        : TASK == RSX_TASK_SSL_SCRIMMAGE ? 2 + 2 * NR
        : TASK == RSX_TASK_SSL_DRIBBLING ? 21 : TASK == RSX_TASK_SSL_CONTESTED ? 14 : 16;
    const int OD = OD_C ? OD_C : P.obs_dim;
#define auxe(ROW) at_byte(bufs.aux, (ix_t)(ROW) * ((ix_t)4 * (ix_t)P.row_stride) + (ix_t)4 * (ix_t)e)   // row ROW of this env in the scalar arena

#ifdef RSX_TIMING
#define RSX_STAMP(i) do { if (lane == 0) bufs.dbg[(size_t)(i) * gridDim.x + blockIdx.x] = __builtin_readcyclecounter(); } while (0)
#else
#define RSX_STAMP(i) do {} while (0)
#endif
#ifdef RSX_TIMING
    if (lane == 0) { sh.dbg = bufs.dbg; bufs.dbg[(size_t)13 * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime(); }  // 100 MHz, chip-wide
#endif
    RSX_STAMP(0);
    // ---- load ----
    Body o; float od, wd, wheels[4];
    const RawBody raw = load_raw<KIND>(P, bufs.state, e, b, is_robot, is_ball);
    // Every launch starts with a cold instruction cache (the first pass through the code costs ~1 k cycles more than the later ones,
    // profiles/r04_timeline_4096.txt).  The VSS single-step kernels touch the 8 KB of their own code that follow the entry point with
    // ONE data load — a lane per 128-byte line; it lands with the state loads, long before the wave gets there — so that those
    // instruction fetches find their lines in the L2: VSS-v0 at 4096 envs 9.02 -> 8.88 us per step (three interleaved rounds).
    // Measured per kernel: later windows (+4, +8, +12 KB) or 16 / 24 KB gain nothing; the SSL kernels lose (11v11 +4 %, 1v6 +0.5 %).
    constexpr bool CODE_PF = KIND == RSX_KIND_VSS && MODE == MODE_STEP;
    uint32_t code_touch = 0;
    if (CODE_PF) {
        unsigned long long pc;
        asm volatile("s_getpc_b64 %0" : "=s"(pc));
        code_touch = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(pc) + 128u * (unsigned)lane));
    }
    int steps = 0; uint32_t episode = 0;
    if (live) {
        steps = __float_as_int(auxe(ROW_STEPS));
        episode = __float_as_uint(auxe(ROW_EPISODE));
    }
    float ou0 = 0.0f, ou1 = 0.0f;
    if (TASK == RSX_TASK_VSS_V0 && is_robot && b >= 1) {
        ou0 = auxe(ROW_OU + 2 * b); ou1 = auxe(ROW_OU + 2 * b + 1);
    }
    float info[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float prev_pot = 0.0f, ep_ret = 0.0f;
    if (is_ball) {
        // VSS-v0: rows 0, 4, 5 (goal counters) are zero except on a terminal step, and the step after it
        // clears them: they are neither read nor — in the common case — written
#pragma unroll
        for (int i = 0; i < ID; ++i)
            if (!(TASK == RSX_TASK_VSS_V0 && (i == 0 || i >= 4))) info[i] = auxe(ROW_INFO + i);
        prev_pot = auxe(ROW_PREV_POT);
        if (TASK != RSX_TASK_VSS_V0) ep_ret = auxe(ROW_EP_RET);   // VSS-v0: derived from the info terms
    }
    // metrics[0] (env-steps) is counted on the device by ONE lane of the grid: launches of a handle
    // are stream-ordered, so a plain read-modify-write is race free and costs no atomic
    const bool counts_steps = (MODE == MODE_STEP || MODE == MODE_ROLLOUT) && blockIdx.x == 0 && lane == 0;
    unsigned long long steps_before = 0;
    if (counts_steps) steps_before = bufs.metrics[0];
    float reward = 0.0f; int term = 0, trunc = 0;
    bool success = false;  // goal scored / course completed / pass received (metrics[2])
    bool against = false;  // goal conceded (metrics[3]; scrimmage)
    bool was_reset = false;
    // caller-fed actions of the agent lane (robot 0); fed launches run a single step
    const bool fed = MODE == MODE_STEP && bufs.actions != nullptr;
    float act[AD];
#pragma unroll
    for (int i = 0; i < AD; ++i) act[i] = 0.0f;
    if (TASK == RSX_TASK_SSL_SCRIMMAGE) {   // every robot is commanded: [B][N][4]
        if (fed && is_robot) {
#pragma unroll
            for (int i = 0; i < AD; ++i) act[i] = at_byte(bufs.actions, (ix_t)4 * (((ix_t)e * (ix_t)N + (ix_t)b) * (ix_t)AD + (ix_t)i));   // ([B][N][AD] is smaller than the state array)
        }
    } else if (fed && is_robot && b == 0) {
#pragma unroll
        for (int i = 0; i < AD; ++i) act[i] = at_byte(bufs.actions, (ix_t)4 * ((ix_t)e * (ix_t)AD + (ix_t)i));
    }

    // placement cache: this body's pose in the env's next episode and the episode id it was made for, loaded with the state
    float pcx = 0.0f, pcy = 0.0f, pcth = 0.0f;
    uint32_t ptag = 0u;
    bool pc_on = false;
    if constexpr (PC) {
        pc_on = bufs.pcache != nullptr;
        if (pc_on && (is_robot || is_ball)) {
            constexpr int NBD = (PC ? NR : 1) + 1;
            const float* const pr = bufs.pcache + (size_t)((tick0 + 1u) & 1u) * (size_t)pcache_rows<NBD>() * B;
            pcx = pr[(size_t)(0 * NBD + b) * B + e]; pcy = pr[(size_t)(1 * NBD + b) * B + e]; pcth = pr[(size_t)(2 * NBD + b) * B + e];
            ptag = __float_as_uint(pr[(size_t)(3 * NBD) * B + e]);
        }
    }

    // single-step launches: this step's random numbers, computed in the shadow of the loads
    StepDraw pre;
    if (MODE == MODE_STEP) pre = draw_for_step<KIND, TASK>(P, env_id, tick0, b, is_robot, fed);

    // All loads land here, once.  Without this the compiler parks a vmcnt(0) at the top of the
    // step loop (loop-carried values come from loads on the first trip), and on gfx9-class
    // counters that wait also drains the previous trip's global STORES: one HBM write round
    // trip per env step in the multi-step (rollout) launches.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    if (CODE_PF) asm volatile("" :: "v"(code_touch));   // (the value itself is of no interest)
    interpret_body<KIND>(raw, is_robot, is_ball, o, od, wd, wheels);
    RSX_STAMP(1);

    for (int it = 0; it < n_steps; ++it) {
        bool ended;
        const float obs_ts = prev_pot;   // the task scalar as this step's observation sees it (before the reward moves it)
        if (mode == 2) {
            const bool flagged = live && bufs.flags[2 * B + e] != 0;   // third row of the flags array: the env mask of rsx_task_reset_to
            if (flagged) {
                episode += 1; steps = 0; ou0 = 0.0f; ou1 = 0.0f;
                if (is_ball) {
#pragma unroll
                    for (int i = 0; i < 10; ++i) info[i] = 0.0f;
#pragma unroll
                    for (int i = 0; i < ID; ++i) auxe(ROW_INFO + i) = 0.0f;
                    ep_ret = 0.0f; prev_pot = 0.0f;
                }
            }
            // (only the re-placed envs: the observation of an env the mask leaves alone stays the one its last step wrote — recomputed here
            // it would see the task scalar AFTER that step's reward moved it, e.g. the checkpoint count of SSLDribbling; found by
            // tests/test_gpu_api_fuzz.py in 3 of 500 sequences)
            if (flagged) write_obs<KIND, TASK>(P, bufs.obs + (size_t)e * OD, b, is_robot, is_ball, o.x, o.y, o.vx, o.vy, o.s, o.c, wd, o.ir, prev_pot);
            wave_sync();
            ended = false;
        } else if (mode == 1) {
            // reset(): nothing to simulate; fall through to the placement block below
            ended = live;
            episode += 1;
            if (is_ball) {
#pragma unroll
                for (int i = 0; i < 10; ++i) info[i] = 0.0f;
#pragma unroll
                for (int i = 0; i < ID; ++i) auxe(ROW_INFO + i) = 0.0f;   // like reset_to above: the info rows of a fresh episode read zero
                ep_ret = 0.0f; prev_pot = 0.0f;
            }
        } else {
            const bool first_step = steps == 0;
            const uint32_t t = tick0 + (uint32_t)it;   // per-step draws are keyed by the handle's step count, not by the env's counters
            if (is_ball && first_step) {
#pragma unroll
                for (int i = 0; i < 10; ++i) info[i] = 0.0f;
                ep_ret = 0.0f;
            }
            const float lastx = o.x, lasty = o.y;  // the reference's last_frame (pre-step)

            // ---- actions -> commands ----
            float q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const StepDraw dr = MODE == MODE_STEP ? pre : draw_for_step<KIND, TASK>(P, env_id, t, b, is_robot, fed);
            if (TASK == RSX_TASK_VSS_V0) {
                if (is_robot) {
                    float a0, a1;
                    if (b == 0) {   // the agent: fed action or the step's uniform draw
                        if (fed) { a0 = act[0]; a1 = act[1]; }
                        else { a0 = dr.v[0]; a1 = dr.v[1]; }
                    } else {  // Ornstein-Uhlenbeck noise, Utils/Utils.py:14-21, on the step's two normals
                        ou0 = (ou0 + P.ou_theta_dt * (0.0f - ou0)) + P.ou_sig_sqdt * dr.v[0];
                        ou1 = (ou1 + P.ou_theta_dt * (0.0f - ou1)) + P.ou_sig_sqdt * dr.v[1];
                        a0 = ou0; a1 = ou1;
                    }
                    q[0] = vss_wheel(a0); q[1] = vss_wheel(a1);
                }
            } else if (TASK == RSX_TASK_SSL_SCRIMMAGE) {  // every robot: (v_x, v_y, v_theta, kick), block b of the step
                if (is_robot) {
                    float a[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = fed ? act[i] : dr.v[i];
                    q[1] = a[0] * T::max_v; q[2] = a[1] * T::max_v; q[3] = a[2] * 10.0f;
                    q[5] = a[3] > 0.9f ? 5.0f : 0.0f;
                }
            } else {  // the SSL tasks: only blue 0 is driven by the agent
                if (is_robot && b == 0) {
                    float a[5] = {0, 0, 0, 0, 0};
#pragma unroll
                    for (int i = 0; i < AD; ++i) a[i] = fed ? act[i] : dr.v[i];
                    ssl_agent_commands<TASK>(a, o.s, o.c, q);
                }
                if (TASK == RSX_TASK_SSL_PASS_ENDURANCE && is_robot && b == 1) q[7] = 1.0f;  // receiver: dribbler on
            }
            if (is_robot) robot_targets<KIND>(P, o, q);

            // ---- physics ----
            RSX_STAMP(2);
            physics<KIND, L, NR>(P, o, b, g, live, sh);
            RSX_STAMP(3);
            if (MODE == MODE_STEP && KIND == RSX_KIND_SSL) {
                // Single-step launches of the SSL tasks: what the rest of the step reads of the parameter block is fetched from the kernarg
                // segment HERE, behind an opaque pointer, instead of being loaded at the kernel's entry and parked in VGPR lanes across
                // the physics (these kernels run out of scalar registers: ~40 v_writelane at entry, ~60 v_readlane after the physics)
                typedef const __attribute__((address_space(4))) uint32_t* kw_t;
                constexpr size_t KOFF = RSX_PARAMS_KERNARG_OFFSET;   // after RSX_HOT_ARGS
                static_assert(sizeof(Params) % 4 == 0 && KOFF % 4 == 0, "parameter block in dwords");
                kw_t pk = (kw_t)__builtin_amdgcn_kernarg_segment_ptr() + KOFF / 4;
                asm volatile("" : "+s"(pk));
                struct Words { uint32_t w[sizeof(Params) / 4]; } raw;
#pragma unroll
                for (size_t i = 0; i < sizeof(Params) / 4; ++i) raw.w[i] = pk[i];
                P = __builtin_bit_cast(Params, raw);
                RSX_UNPACK_HOT(P);
            }

            // ---- wire-format values, observation, reward ----
            if (is_robot) {
                od = o.th; wd = o.om * K::rad2deg;
                if (KIND == RSX_KIND_SSL) wheel_speeds<KIND>(P, o, wheels);
                // omega lives in HBM as deg/s: keep the lane's copy equal to what a reload gives.
                o.om = wd * K::deg2rad;
                // the sub-steps carried (c, s) by small rotations; re-derive them exactly from the
                // stored heading: this is what the observation reports and what a reload (the next
                // launch, or the next step of a multi-step launch) starts from
                sincos_f32(o.th * K::deg2rad, o.s, o.c);
            } else if (is_ball) {
                o.z = (K::r_ball + o.z) - K::r_ball;  // height goes through the wire format too
            }
            write_obs<KIND, TASK>(P, bufs.obs + (size_t)e * OD, b, is_robot, is_ball, o.x, o.y, o.vx, o.vy, o.s, o.c, wd, o.ir, obs_ts);
            // what the reward lane (the ball's) needs from the robots' lanes
            if (is_robot && b == 0) {
                float* xr = sh.x0[g];
                xr[0] = o.x; xr[1] = o.y;
                if (TASK == RSX_TASK_VSS_V0) { xr[2] = o.vx; xr[3] = o.vy; xr[4] = q[0]; xr[5] = q[1]; }
                else if (TASK == RSX_TASK_SSL_STATIC_DEFENDERS || TASK == RSX_TASK_SSL_CONTESTED) {
                    xr[6] = lastx; xr[7] = lasty;
                    xr[8] = wheels[0]; xr[9] = wheels[1]; xr[10] = wheels[2]; xr[11] = wheels[3];
                }
            } else if (is_robot) {
                float* xr = sh.x0[g];
                if (TASK == RSX_TASK_SSL_DRIBBLING) xr[1 + b] = (fabsf(o.vx) > 0.05f || fabsf(o.vy) > 0.05f) ? 1.0f : 0.0f;
                if (TASK == RSX_TASK_SSL_CONTESTED && b == 1) xr[2] = (fabsf(o.vx) > 0.1f || fabsf(o.vy) > 0.1f) ? 1.0f : 0.0f;
                if (TASK == RSX_TASK_SSL_PASS_ENDURANCE && b == 1) { xr[2] = o.x; xr[3] = o.y; xr[4] = o.ir ? 1.0f : 0.0f; }
            }
            wave_sync();
            if (is_ball) {
                const float* xr = sh.x0[g];
                task_reward<KIND, TASK>(P, xr, o.x, o.y, lastx, lasty, first_step, prev_pot, info, reward, term, success, against);
                ep_ret = ep_ret + reward;
            }
            steps += 1;
            trunc = steps >= P.max_steps;
            // episode-end flag of the env: held by its ball lane (lane N*G + g), spread with one
            // ballot instead of an LDS round trip
            const unsigned long long endm = __ballot(is_ball && (term | trunc));
            ended = live && ((endm >> (LaneMap<L>::slot(N, g))) & 1ull) != 0;
            if (is_ball) {
                // info is reported as it stands after this step (cleared lazily at the next
                // episode's first step), like the dict the reference returns with `done`
#pragma unroll
                for (int i = 0; i < ID; ++i)
                    if (!(TASK == RSX_TASK_VSS_V0 && (i == 0 || i >= 4)) || term || first_step) auxe(ROW_INFO + i) = info[i];
                auxe(ROW_REWARD) = reward;
                if (MODE == MODE_STEP) { bufs.flags[(ix_t)e] = (uint8_t)term; bufs.flags[(ix_t)P.num_envs + (ix_t)e] = (uint8_t)trunc; }
                else { bufs.flags[e] = (uint8_t)term; bufs.flags[B + e] = (uint8_t)trunc; }
            }
        }

        RSX_STAMP(4);
        // ---- episode end: same-step auto-reset (or reset()) ----
        if (RSX_RARE_B(KIND, 4, __any(ended))) {
            if (ended && mode == 0) {  // terminal observation
                write_obs<KIND, TASK>(P, bufs.final_obs + (size_t)e * OD, b, is_robot, is_ball, o.x, o.y, o.vx, o.vy, o.s, o.c, wd, o.ir, obs_ts);
            }
            if (ended && mode == 0) episode += 1;   // every lane of the env: the new episode's id
            if (KIND == RSX_KIND_VSS) {
                if (ended && is_ball && mode == 0) {
                    unsigned long long* const ms = metric_slot(bufs);
                    atomicAdd(&ms[1], 1ull);
                    if (info[4] > 0.0f) atomicAdd(&ms[2], 1ull);
                    if (info[5] > 0.0f) atomicAdd(&ms[3], 1ull);
                    atomicAdd(&ms[4], (unsigned long long)__float2ll_rn(vss_episode_return(info) * 1048576.0f));
                    atomicAdd(&ms[5], (unsigned long long)steps);
                    if (trunc && !term) atomicAdd(&ms[6], 1ull);
                }
            } else if (mode == 0) {
                // SSL tasks (short episodes: several resetting waves in every launch): the ball lane
                // holds the increments, lanes 0..5 of the env add one each, so the wave issues ONE
                // atomic instruction instead of six guarded ones, each wrapped in wave-reduction
                // code by the compiler.  Measured: static defenders 11.33 -> 11.03 us; VSS-v0 does
                // not gain (8.83 -> 8.88) and keeps the plain form.
                static_assert(KIND == RSX_KIND_VSS || L >= 6, "metrics fan-out needs 6 lanes per env");
                uint32_t* const mv = reinterpret_cast<uint32_t*>(sh.x0[g]);   // 12 words, free again after the reward
                if (ended && is_ball) {
                    unsigned long long inc[6];
                    inc[0] = 1ull;
                    inc[1] = success ? 1ull : 0ull;
                    inc[2] = against ? 1ull : 0ull;
                    inc[3] = (unsigned long long)__float2ll_rn(ep_ret * 1048576.0f);
                    inc[4] = (unsigned long long)steps;
                    inc[5] = (trunc && !term) ? 1ull : 0ull;
#pragma unroll
                    for (int k = 0; k < 6; ++k) { mv[2 * k] = (uint32_t)inc[k]; mv[2 * k + 1] = (uint32_t)(inc[k] >> 32); }
                }
                wave_sync();
                if (ended && b < 6) {
                    const unsigned long long v = (unsigned long long)mv[2 * b] | ((unsigned long long)mv[2 * b + 1] << 32);
                    if (v) atomicAdd(&metric_slot(bufs)[1 + b], v);
                }
            }
            RSX_STAMP(15);
            // an env whose next episode's poses were in the cache takes them from there; the others are placed here
            const bool hit = PC && pc_on && ended && ptag == episode;
            const bool place = ended && !hit;
            const bool any_place = !PC || __any(place);
            if (any_place) {
                if (place) place_predraw<TASK, L>(P, env_id, episode, b, sh.draws[g]);
                wave_sync();  // draws published; stage rows of ended envs are about to be overwritten
            }
            RSX_STAMP(16);
            float4 pz = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            // A single-step launch waits for its slowest wave, which is one that resets an env:
            // there the placement runs in its parallel form.  A multi-step launch pays for the
            // average wave instead, and the sequential form issues fewer instructions.
            if (TASK == RSX_TASK_SSL_SCRIMMAGE) {
                // robot k in cell (k % 6, k / 6) of a 6 x 4 grid, one Philox block per body: every lane places itself
                if (ended && (is_robot || is_ball)) {
                    const u32x4 u = philox4x32(env_id, episode, (uint32_t)b, DOM_PLACE, P.key0, P.key1);
                    const float jx = u01(u.x) * 2.0f - 1.0f, jy = u01(u.y) * 2.0f - 1.0f;
                    if (is_ball) pz = make_float4(P.sc_jb * jx, P.sc_jb * jy, 0.0f, 0.0f);
                    else pz = make_float4(P.sc_sx * ((float)(b % 6) - 2.5f) + P.sc_j * jx,
                                          P.sc_sy * ((float)(b / 6) - 1.5f) + P.sc_j * jy, 360.0f * u01(u.z), 0.0f);
                }
            } else if ((TASK == RSX_TASK_VSS_V0 || TASK == RSX_TASK_SSL_STATIC_DEFENDERS) && MODE != MODE_ROLLOUT) {
                if (any_place && place) pz = place_env_parallel<TASK, L, NR>(P, N, env_id, episode, b, g, is_robot, sh.A, sh.draws[g]);
                if (hit) pz = make_float4(pcx, pcy, pcth, 0.0f);
                if (PC && pc_on && bufs.pcstats && ended && is_ball) atomicAdd(&bufs.pcstats[hit ? 0 : 1], 1ull);
            } else {
                if (ended && is_ball) place_env<TASK, L>(P, N, env_id, episode, g, sh.A, sh.draws[g]);
                wave_sync();
                if (ended && (is_robot || is_ball)) pz = sh.A[LaneMap<L>::slot(b, g)];
            }
            RSX_STAMP(17);
            if (ended) {
                steps = 0; ou0 = 0.0f; ou1 = 0.0f; was_reset = true;
                if (TASK >= RSX_TASK_SSL_DRIBBLING) prev_pot = 0.0f;  // checkpoints_count / stopped_steps
                if (is_robot || is_ball) {
                    o = Body{};
                    o.x = pz.x; o.y = pz.y;
                    od = pz.z; wd = 0.0f;
                    wheels[0] = wheels[1] = wheels[2] = wheels[3] = 0.0f;
                    if (is_robot) { o.th = od; sincos_f32(o.th * K::deg2rad, o.s, o.c); }
                }
                write_obs<KIND, TASK>(P, bufs.obs + (size_t)e * OD, b, is_robot, is_ball, o.x, o.y, o.vx, o.vy, o.s, o.c, wd, 0, 0.0f);
            }
            wave_sync();
            RSX_STAMP(18);
        }

        wave_sync();
    }

    RSX_STAMP(5);
    // ---- store (wire format: degrees, deg/s; SSL: infrared + wheel speeds) ----
    if (mode != 2) store_body<KIND>(P, bufs.state, e, b, is_robot, is_ball, o, od, wd, wheels, P.n_sub != 0 || was_reset);
    if (live && b == 0) {
        auxe(ROW_STEPS) = __int_as_float(steps);
        auxe(ROW_EPISODE) = __uint_as_float(episode);
    }
    if (TASK == RSX_TASK_VSS_V0 && is_robot && b >= 1) {
        auxe(ROW_OU + 2 * b) = ou0; auxe(ROW_OU + 2 * b + 1) = ou1;
    }
    if (is_ball) {
        auxe(ROW_PREV_POT) = prev_pot;
        if (TASK != RSX_TASK_VSS_V0) auxe(ROW_EP_RET) = ep_ret;
#undef auxe
    }
    if (counts_steps) bufs.metrics[0] = steps_before + (unsigned long long)P.num_envs * (unsigned long long)n_steps;
    RSX_STAMP(6);
#ifdef RSX_TIMING
    __builtin_amdgcn_s_waitcnt(0x0F70);
    RSX_STAMP(7);
    if (lane == 0) bufs.dbg[(size_t)14 * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace rsx
