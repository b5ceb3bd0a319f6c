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
//     envs.  lane = body * G + env_in_wave, so the G lanes that own "body k" of neighbouring
//     envs are adjacent and read adjacent floats of SoA row k: every row access of a wave is
//     G*4 contiguous bytes per body (32 B segments at L = 8), and the 6..11 rows of one body
//     share cache lines with the neighbouring tiles handled by the SAME XCD (tile -> XCD map
//     below), so every byte fetched into an L2 is used.
//   * the all-pairs contact test is a Jacobi sweep: each lane publishes (x, y, vx, vy) as one
//     float4 in LDS and reads the other bodies' float4 back (ds_read_b128, broadcast inside a
//     group, G distinct 16-B slots per instruction -> conflict free).  All sub-steps of one
//     env.step() run out of registers + LDS; HBM is touched once for the load and once for the
//     store of the SoA state.
//   * SSL: the robot lane evaluates its own robot-ball contact (kicker-mouth geometry) and
//     publishes the ball-side impulse / dribbler / kick record; the ball lane only sums the
//     N records in index order.  No lane re-derives another lane's geometry.
//   * 64-thread workgroups (one wave): a 4096-env VSS batch is 512 workgroups, two per CU,
//     so all 256 CUs work; block b runs on XCD b % 8 and is mapped to tile
//     (b % 8) * tiles_per_xcd + b / 8 so each XCD's L2 sees one contiguous 1/8 of every row.
//   * fp32 throughout, -ffp-contract=off, fixed summation order (body index order) -> results
//     are bit-identical for any batch size, position in the batch, L, or shard.
#pragma once
#include <hip/hip_runtime.h>

#include "rsx_math.hpp"
#include "rsx_params.hpp"

namespace rsx {

struct Buffers {
    float* state;          // [state_dim+1][B]
    const float* cmds;     // [N*C][B]          (raw simulator path)
    const float* actions;  // [B][act_dim] or nullptr = random
    float* obs;            // [B][obs_dim]
    float* reward;         // [B]
    uint8_t* terminated;   // [B]
    uint8_t* truncated;    // [B]
    float* info;           // [info_dim][B]
    float* final_obs;      // [B][obs_dim]
    int* steps;            // [B]
    uint32_t* episode;     // [B]
    float* ou;             // [2*N][B]
    float* prev_pot;       // [B]
    float* ep_ret;         // [B]
    unsigned long long* metrics;  // [RSX_METRICS]
};

// what one lane keeps in registers for its body
struct Body {
    float x, y, vx, vy;       // all bodies
    float th, om, c, s;       // robots: heading (rad), rate (rad/s), cos/sin(heading)
    float t0, t1, t2;         // VSS: v target, omega target | SSL: local vx, vy, omega targets
    float kick_x, kick_z;     // SSL
    float z, vz;              // ball: height above rest, vertical speed
    int drib, ir;
};

template <int L>
struct Shared {
    float4 A[64];   // x, y, vx, vy of every body (slot = lane)
    float4 Bq[64];  // SSL robot -> ball record 0: dvx, dvy, dpx, dpy (ball side)
    float4 Cq[64];  // SSL robot -> ball record 1: flags, ovx, ovy, ovz
    float zb[64 / L];          // ball height per env
    float x0[64 / L][12];      // robot 0 -> reward lane exchange
    int flag[64 / L];          // episode-end flag per env
    float stage[(64 / L) * 64];  // obs staging, [env][obs_dim], obs_dim <= 64
};

// clamp a circle (radius r, restitution rest) into the playable region
template <int KIND>
__device__ __forceinline__ void walls(const Params& P, float r, float rest, float& x, float& y,
                                      float& vx, float& vy) {
    float ax = fabsf(x), ay = fabsf(y);
    float sx = x < 0.0f ? -1.0f : 1.0f, sy = y < 0.0f ? -1.0f : 1.0f;
    if (KIND == RSX_KIND_VSS) {
        if (ax > P.half_len) {  // centre inside a goal box
            float yl = P.ghw - r;
            if (ay > yl) { y = sy * yl; if (vy * sy > 0.0f) vy = -rest * vy; }
            float xl = (P.half_len + P.gd) - r;
            if (ax > xl) { x = sx * xl; if (vx * sx > 0.0f) vx = -rest * vx; }
        } else {
            float yl = P.half_wid - r;
            if (ay > yl) { y = sy * yl; if (vy * sy > 0.0f) vy = -rest * vy; }
            float xl = P.half_len - r;
            if (ax > xl && ay > P.ghw - r) { x = sx * xl; if (vx * sx > 0.0f) vx = -rest * vx; }
        }
    } else {
        float yl = (P.half_wid + P.margin) - r;
        if (ay > yl) { y = sy * yl; if (vy * sy > 0.0f) vy = -rest * vy; ay = yl; }
        float xl = (P.half_len + P.margin) - r;
        if (ax > xl) { x = sx * xl; if (vx * sx > 0.0f) vx = -rest * vx; ax = xl; }
        if (ax > P.half_len) {
            float back = P.half_len + P.gd;
            if (ay < P.ghw) {
                if (ax < back) {  // inside the goal
                    if (ax > back - r) { x = sx * (back - r); if (vx * sx > 0.0f) vx = -rest * vx; }
                    if (ay > P.ghw - r) { y = sy * (P.ghw - r); if (vy * sy > 0.0f) vy = -rest * vy; }
                } else if (ax < back + r) {  // behind the back wall
                    x = sx * (back + r); if (vx * sx < 0.0f) vx = -rest * vx;
                }
            } else if (ay < P.ghw + r && ax < back) {  // outside, touching a side wall
                y = sy * (P.ghw + r); if (vy * sy < 0.0f) vy = -rest * vy;
            }
        }
    }
}

// per-step command processing of a robot lane: wheel / velocity commands -> targets
template <int KIND>
__device__ __forceinline__ void robot_targets(const Params& P, Body& o, const float* q /*C cmds*/) {
    if (KIND == RSX_KIND_VSS) {
        float wl = clampf(q[0], -P.w_max, P.w_max);
        float wr = clampf(q[1], -P.w_max, P.w_max);
        o.t0 = (wl + wr) * P.half_rw;
        o.t1 = (wr - wl) * P.rw_2b;
        o.t2 = 0.0f; o.kick_x = 0.0f; o.kick_z = 0.0f; o.drib = 0;
    } else {
        float vtx, vty, omt;
        if (q[0] != 0.0f) {
            float w0 = clampf(q[1], -P.w_max, P.w_max), w1 = clampf(q[2], -P.w_max, P.w_max);
            float w2 = clampf(q[3], -P.w_max, P.w_max), w3 = clampf(q[4], -P.w_max, P.w_max);
            vtx = (((P.pinv[0][0] * w0 + P.pinv[0][1] * w1) + P.pinv[0][2] * w2) + P.pinv[0][3] * w3) * P.r_wheel;
            vty = (((P.pinv[1][0] * w0 + P.pinv[1][1] * w1) + P.pinv[1][2] * w2) + P.pinv[1][3] * w3) * P.r_wheel;
            omt = (((P.pinv[2][0] * w0 + P.pinv[2][1] * w1) + P.pinv[2][2] * w2) + P.pinv[2][3] * w3) * P.r_wheel;
        } else {
            vtx = q[1]; vty = q[2]; omt = q[3];
            float m = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float wi = ((vty * P.wc[i] - vtx * P.ws[i]) + omt * P.r_robot) * P.inv_rw;
                float a = fabsf(wi);
                if (a > m) m = a;
            }
            if (m > P.w_max) { float sc = P.w_max / m; vtx = vtx * sc; vty = vty * sc; omt = omt * sc; }
        }
        o.t0 = vtx; o.t1 = vty; o.t2 = omt;
        o.kick_x = q[5]; o.kick_z = q[6]; o.drib = q[7] != 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// n_sub sub-steps of one env.step() for the body held by this lane.
//   b = body index of the lane (0..N-1 robots, N ball, > N idle), g = env slot in the wave
// ---------------------------------------------------------------------------------------------
template <int KIND, int L>
__device__ __forceinline__ void physics(const Params& P, Body& o, const int b, const int g,
                                        const bool live, Shared<L>& sh) {
    constexpr int G = 64 / L;
    const int N = P.n_robots;
    const bool is_robot = live && b < N;
    const bool is_ball = live && b == N;
    const int lane = b * G + g;

    for (int sub = 0; sub < P.n_sub; ++sub) {
        // ---- A: actuation + integration ----
        if (is_robot) {
            float vf = o.vx * o.c + o.vy * o.s;
            float vl = o.vy * o.c - o.vx * o.s;
            if (KIND == RSX_KIND_VSS) {
                vf = vf + clampf(o.t0 - vf, -P.a_lin_h, P.a_lin_h);
                vl = vl - clampf(vl, -P.a_lat_h, P.a_lat_h);
                o.om = o.om + clampf(o.t1 - o.om, -P.a_ang_h, P.a_ang_h);
            } else {
                float dx = o.t0 - vf, dy = o.t1 - vl;
                float d2 = dx * dx + dy * dy;
                if (d2 > P.a_lin_h2) { float sc = P.a_lin_h / sqrtf(d2); dx = dx * sc; dy = dy * sc; }
                vf = vf + dx; vl = vl + dy;
                o.om = o.om + clampf(o.t2 - o.om, -P.a_ang_h, P.a_ang_h);
            }
            o.vx = vf * o.c - vl * o.s;
            o.vy = vf * o.s + vl * o.c;
            o.x = o.x + o.vx * P.h;
            o.y = o.y + o.vy * P.h;
            o.th = o.th + o.om * P.h;
            if (o.th > P.pi) o.th = o.th - P.two_pi;
            else if (o.th < -P.pi) o.th = o.th + P.two_pi;
            sincos_f32(o.th, o.s, o.c);
        } else if (is_ball) {
            if (o.z > 0.0f || o.vz > 0.0f) {
                o.vz = o.vz - P.g_h;
                o.z = o.z + o.vz * P.h;
                if (o.z <= 0.0f) {
                    o.z = 0.0f;
                    o.vz = -o.vz * P.e_ground;
                    if (o.vz < P.vz_min) o.vz = 0.0f;
                }
            } else {
                float sp2 = o.vx * o.vx + o.vy * o.vy;
                if (sp2 > 0.0f) {
                    float sp = sqrtf(sp2), ns = sp - P.mu_g_h;
                    if (ns < 0.0f) ns = 0.0f;
                    float k = ns / sp;
                    o.vx = o.vx * k; o.vy = o.vy * k;
                }
            }
            o.x = o.x + o.vx * P.h;
            o.y = o.y + o.vy * P.h;
        }

        // ---- B: contacts, Jacobi over the post-integration snapshot ----
        sh.A[lane] = make_float4(o.x, o.y, o.vx, o.vy);
        if (is_ball) sh.zb[g] = o.z;
        __syncthreads();
        const bool ball_low = sh.zb[g] < P.robot_h;
        float avx = 0.0f, avy = 0.0f, apx = 0.0f, apy = 0.0f;

        if (KIND == RSX_KIND_VSS) {
            // every pair is circle-circle; only the constants depend on the pair type
            if (is_robot || is_ball) {
                for (int j = 0; j <= N; ++j) {
                    if (j == b) continue;
                    const float4 oj = sh.A[j * G + g];
                    const bool rb = is_ball || j == N;
                    const float rs2 = rb ? P.rs_rb2 : P.rs_rr2;
                    float dx = oj.x - o.x, dy = oj.y - o.y;
                    float d2 = dx * dx + dy * dy;
                    if (d2 < rs2 && d2 > 0.0f && (!rb || ball_low)) {
                        const float rs = rb ? P.rs_rb : P.rs_rr;
                        const float ope = rb ? P.ope_rb : P.ope_rr;
                        const float w = is_ball ? P.w_rb_b : (j == N ? P.w_rb_r : P.w_rr);
                        float d = sqrtf(d2), inv = 1.0f / d;
                        float nx = dx * inv, ny = dy * inv, pen = rs - d;
                        float vn = (oj.z - o.vx) * nx + (oj.w - o.vy) * ny;
                        if (vn < 0.0f) { float q = ope * vn * w; avx = avx + q * nx; avy = avy + q * ny; }
                        float pc = P.beta * pen * w;
                        apx = apx - pc * nx; apy = apy - pc * ny;
                    }
                }
            }
            o.vx = o.vx + avx; o.vy = o.vy + avy;
            o.x = o.x + apx; o.y = o.y + apy;
        } else {
            bool ovr = false, okick = false;
            float ovx = 0.0f, ovy = 0.0f, ovz = 0.0f;
            if (is_robot) {
                for (int j = 0; j < N; ++j) {  // robot - robot
                    if (j == b) continue;
                    const float4 oj = sh.A[j * G + g];
                    float dx = oj.x - o.x, dy = oj.y - o.y;
                    float d2 = dx * dx + dy * dy;
                    if (d2 < P.rs_rr2 && d2 > 0.0f) {
                        float d = sqrtf(d2), inv = 1.0f / d;
                        float nx = dx * inv, ny = dy * inv, pen = P.rs_rr - d;
                        float vn = (oj.z - o.vx) * nx + (oj.w - o.vy) * ny;
                        if (vn < 0.0f) { float q = P.ope_rr * vn * P.w_rr; avx = avx + q * nx; avy = avy + q * ny; }
                        float pc = P.beta * pen * P.w_rr;
                        apx = apx - pc * nx; apy = apy - pc * ny;
                    }
                }
                // robot - ball: kicker mouth (flat face at dck) or body circle; n points robot -> ball
                const float4 ob = sh.A[N * G + g];
                float dx = ob.x - o.x, dy = ob.y - o.y;
                float nx = 0.0f, ny = 0.0f, pen = -1.0f;
                bool mouth = false, touch = false;
                if (ball_low) {
                    float lx = dx * o.c + dy * o.s, ly = dy * o.c - dx * o.s;
                    if (fabsf(ly) < P.half_kw && lx > 0.0f) {
                        mouth = true; pen = P.dck_rb - lx; nx = o.c; ny = o.s; touch = pen > 0.0f;
                    } else {
                        float d2 = dx * dx + dy * dy;
                        if (d2 < P.rs_rb2 && d2 > 0.0f) {
                            float d = sqrtf(d2), inv = 1.0f / d;
                            nx = dx * inv; ny = dy * inv; pen = P.rs_rb - d; touch = true;
                        }
                    }
                }
                float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                int fl = 0;
                if (touch) {
                    float vn = (ob.z - o.vx) * nx + (ob.w - o.vy) * ny;
                    if (vn < 0.0f) {
                        float q = P.ope_rb * vn * P.w_rb_r; avx = avx + q * nx; avy = avy + q * ny;
                        float qb = P.ope_rb * vn * P.w_rb_b; r0.x = qb * nx; r0.y = qb * ny; fl |= 1;
                    }
                    float pc = P.beta * pen * P.w_rb_r;
                    apx = apx - pc * nx; apy = apy - pc * ny;
                    float pb = P.beta * pen * P.w_rb_b; r0.z = pb * nx; r0.w = pb * ny; fl |= 2;
                }
                o.ir = mouth && pen > -P.ir_tol;
                if (o.ir) {  // infrared: kicker / dribbler act on the ball
                    if (o.kick_x > 0.0f || o.kick_z > 0.0f) {
                        fl |= 4 | 8;
                        r1.y = o.vx + o.kick_x * o.c; r1.z = o.vy + o.kick_x * o.s; r1.w = o.kick_z;
                    } else if (o.drib) {
                        float hx = o.x + P.dck_rb * o.c, hy = o.y + P.dck_rb * o.s;
                        float cvx = (hx - ob.x) * P.drib_gain, cvy = (hy - ob.y) * P.drib_gain;
                        float m2 = cvx * cvx + cvy * cvy;
                        if (m2 > P.drib_vmax2) { float sc = P.drib_vmax / sqrtf(m2); cvx = cvx * sc; cvy = cvy * sc; }
                        fl |= 4;
                        r1.y = (o.vx - o.om * P.dck_rb * o.s) + cvx;
                        r1.z = (o.vy + o.om * P.dck_rb * o.c) + cvy;
                    }
                }
                r1.x = __int_as_float(fl);
                sh.Bq[lane] = r0; sh.Cq[lane] = r1;
            }
            __syncthreads();
            if (is_ball) {
                for (int j = 0; j < N; ++j) {  // ball: sum the robots' records in index order
                    const float4 r1 = sh.Cq[j * G + g];
                    const int fl = __float_as_int(r1.x);
                    if (fl) {
                        const float4 r0 = sh.Bq[j * G + g];
                        if (fl & 1) { avx = avx - r0.x; avy = avy - r0.y; }
                        if (fl & 2) { apx = apx + r0.z; apy = apy + r0.w; }
                        if (fl & 4) { ovr = true; okick = (fl & 8) != 0; ovx = r1.y; ovy = r1.z; ovz = r1.w; }
                    }
                }
            }
            o.vx = o.vx + avx; o.vy = o.vy + avy;
            o.x = o.x + apx; o.y = o.y + apy;
            if (ovr) {
                o.vx = ovx; o.vy = ovy;
                if (okick && ovz > 0.0f) o.vz = ovz;
            }
        }

        // ---- C: walls ----
        if (is_robot) walls<KIND>(P, P.r_robot, P.e_wr, o.x, o.y, o.vx, o.vy);
        else if (is_ball) walls<KIND>(P, P.r_ball, P.e_wb, o.x, o.y, o.vx, o.vy);
        __syncthreads();  // A / Bq / Cq are rewritten by the next sub-step
    }
}

// ---------------------------------------------------------------------------------------------
// SoA state access
// ---------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void load_body(const Params& P, const float* __restrict__ st, int e,
                                          int b, bool is_robot, bool is_ball, Body& o,
                                          float& th_deg) {
    const size_t B = (size_t)P.num_envs;
    o = Body{};
    th_deg = 0.0f;
    if (is_robot) {
        const float* r = st + (size_t)(5 + P.rs * b) * B + e;
        o.x = r[0]; o.y = r[B]; th_deg = r[2 * B]; o.vx = r[3 * B]; o.vy = r[4 * B];
        float om_deg = r[5 * B];
        if (KIND == RSX_KIND_SSL) o.ir = r[6 * B] != 0.0f;
        o.th = th_deg * P.deg2rad;
        o.om = om_deg * P.deg2rad;
        sincos_f32(o.th, o.s, o.c);
    } else if (is_ball) {
        const float* r = st + e;
        o.x = r[0]; o.y = r[B]; o.z = r[2 * B] - P.r_ball; o.vx = r[3 * B]; o.vy = r[4 * B];
        o.vz = r[(size_t)P.state_dim * B];
    }
}

// wire-format values of a body after the step: returns theta/omega in degrees, wheel speeds
template <int KIND>
__device__ __forceinline__ void store_body(const Params& P, float* __restrict__ st, int e, int b,
                                           bool is_robot, bool is_ball, const Body& o,
                                           float& th_deg, float& om_deg, float w[4]) {
    const size_t B = (size_t)P.num_envs;
    if (is_robot) {
        float* r = st + (size_t)(5 + P.rs * b) * B + e;
        th_deg = o.th * P.rad2deg; om_deg = o.om * P.rad2deg;
        r[0] = o.x; r[B] = o.y; r[2 * B] = th_deg; r[3 * B] = o.vx; r[4 * B] = o.vy; r[5 * B] = om_deg;
        if (KIND == RSX_KIND_SSL) {
            if (P.n_sub) r[6 * B] = o.ir ? 1.0f : 0.0f;
            float vf = o.vx * o.c + o.vy * o.s;
            float vl = o.vy * o.c - o.vx * o.s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i] = ((vl * P.wc[i] - vf * P.ws[i]) + o.om * P.r_robot) * P.inv_rw;
                r[(7 + i) * B] = w[i];
            }
        }
    } else if (is_ball) {
        float* r = st + e;
        r[0] = o.x; r[B] = o.y; r[2 * B] = P.r_ball + o.z; r[3 * B] = o.vx; r[4 * B] = o.vy;
        r[(size_t)P.state_dim * B] = o.vz;
    }
}

// block -> tile map: block b runs on XCD b % 8 (observed dispatch order; used for L2 affinity
// only, never for correctness), so give each XCD one contiguous range of tiles.
__device__ __forceinline__ int tile_of_block(int nblocks) {
    const int per = nblocks >> 3;  // grid is a multiple of 8
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

// =============================================================================================
// raw simulator step: robosim.step(cmds) + get_state() on the SoA buffers
// =============================================================================================
template <int KIND, int L>
__global__ __launch_bounds__(64) void sim_step_kernel(const Params P, const Buffers bufs) {
    constexpr int G = 64 / L;
    __shared__ Shared<L> sh;
    const int lane = threadIdx.x;
    const int b = lane / G, g = lane % G;
    const int tile = tile_of_block(gridDim.x);
    const int e = tile * G + g;
    const int N = P.n_robots;
    const bool live = e < P.num_envs;
    const bool is_robot = live && b < N, is_ball = live && b == N;
    const size_t B = (size_t)P.num_envs;

    Body o; float th_deg;
    load_body<KIND>(P, bufs.state, e, b, is_robot, is_ball, o, th_deg);
    if (is_robot) {
        float q[8];
        const float* c = bufs.cmds + (size_t)(b * P.cmd_dim) * B + e;
#pragma unroll
        for (int i = 0; i < (KIND == RSX_KIND_VSS ? 2 : 8); ++i) q[i] = c[i * B];
        robot_targets<KIND>(P, o, q);
    }
    physics<KIND, L>(P, o, b, g, live, sh);
    float od, wd, w[4];
    store_body<KIND>(P, bufs.state, e, b, is_robot, is_ball, o, od, wd, w);
}

// =============================================================================================
// fused task step
// =============================================================================================

// observation entries owned by this lane -> staging row of its env
// (vss_gym.py:93-117, static_defenders.py:90-112); values are the WIRE-format state.
template <int TASK>
__device__ __forceinline__ void write_obs(const Params& P, float* __restrict__ row, int b,
                                          bool is_robot, bool is_ball, float x, float y, float vx,
                                          float vy, float th_deg, float om_deg, int ir) {
    const float lo = -1.2f, hi = 1.2f;
    if (is_ball) {
        row[0] = clampf(x * P.inv_max_pos, lo, hi);
        row[1] = clampf(y * P.inv_max_pos, lo, hi);
        row[2] = clampf(vx * P.inv_max_v, lo, hi);
        row[3] = clampf(vy * P.inv_max_v, lo, hi);
    } else if (is_robot) {
        if (b < P.n_blue) {
            constexpr int W = TASK == RSX_TASK_VSS_V0 ? 7 : 8;
            float* r = row + 4 + W * b;
            float sn, cs;
            sincos_f32(th_deg * P.deg2rad, sn, cs);
            r[0] = clampf(x * P.inv_max_pos, lo, hi);
            r[1] = clampf(y * P.inv_max_pos, lo, hi);
            r[2] = sn; r[3] = cs;
            r[4] = clampf(vx * P.inv_max_v, lo, hi);
            r[5] = clampf(vy * P.inv_max_v, lo, hi);
            r[6] = clampf(om_deg * P.inv_max_w, lo, hi);
            if (TASK == RSX_TASK_SSL_STATIC_DEFENDERS) r[7] = ir ? 1.0f : 0.0f;
        } else {
            constexpr int WB = TASK == RSX_TASK_VSS_V0 ? 7 : 8;
            constexpr int WY = TASK == RSX_TASK_VSS_V0 ? 5 : 2;
            float* r = row + 4 + WB * P.n_blue + WY * (b - P.n_blue);
            r[0] = clampf(x * P.inv_max_pos, lo, hi);
            r[1] = clampf(y * P.inv_max_pos, lo, hi);
            if (TASK == RSX_TASK_VSS_V0) {
                r[2] = clampf(vx * P.inv_max_v, lo, hi);
                r[3] = clampf(vy * P.inv_max_v, lo, hi);
                r[4] = clampf(om_deg * P.inv_max_w, lo, hi);
            }
        }
    }
}

// vss_gym.py:235-254
__device__ __forceinline__ float vss_wheel(const Params& P, float a) {
    float v = a * P.max_v;
    v = clampf(v, -P.max_v, P.max_v);
    if (-P.deadzone < v && v < P.deadzone) v = 0.0f;
    return v * P.inv_rw;
}

// Random placement of one env (vss_gym.py:194-233 / static_defenders.py:214-254 with Philox
// draws).  Runs on the env's ball lane; poses go to sh.A[body slot] = (x, y, theta_deg, 0).
template <int TASK, int L>
__device__ __forceinline__ void place_env(const Params& P, uint32_t env_id, uint32_t episode, int g,
                                       Shared<L>& sh) {
    constexpr int G = 64 / L;
    const int N = P.n_robots;
    uint32_t n = 0;
    int first = 0;
    float bx, by;
    if (TASK == RSX_TASK_SSL_STATIC_DEFENDERS) {
        bx = 0.0f; by = 0.0f;
        for (int t = 0; t < 64; ++t) {
            u32x4 u = philox4x32_10(env_id, episode, n++, DOM_PLACE, P.key0, P.key1);
            bx = P.pl_xlo + P.pl_xspan * u01(u.x);
            by = P.pl_ylo + P.pl_yspan * u01(u.y);
            if (!(bx > P.pen_x && fabsf(by) < P.half_pen_wid)) break;
        }
        sh.A[0 * G + g] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // blue 0 at the origin
        first = 1;
    } else {
        u32x4 u = philox4x32_10(env_id, episode, n++, DOM_PLACE, P.key0, P.key1);
        bx = P.pl_xlo + P.pl_xspan * u01(u.x);
        by = P.pl_ylo + P.pl_yspan * u01(u.y);
    }
    sh.A[N * G + g] = make_float4(bx, by, 0.0f, 0.0f);
    for (int k = first; k < N; ++k) {
        float x = 0.0f, y = 0.0f;
        for (int t = 0; t < 64; ++t) {
            u32x4 u = philox4x32_10(env_id, episode, n++, DOM_PLACE, P.key0, P.key1);
            x = P.pl_xlo + P.pl_xspan * u01(u.x);
            y = P.pl_ylo + P.pl_yspan * u01(u.y);
            bool ok = true;
            {   // ball first, then (static defenders) blue 0, then the robots placed so far
                float dx = x - bx, dy = y - by;
                if (dx * dx + dy * dy < P.pl_min_d2) ok = false;
            }
            for (int q = 0; q < k; ++q) {
                const float4 pq = sh.A[q * G + g];
                float dx = x - pq.x, dy = y - pq.y;
                if (dx * dx + dy * dy < P.pl_min_d2) ok = false;
            }
            if (ok) break;
        }
        u32x4 u = philox4x32_10(env_id, episode, n++, DOM_PLACE, P.key0, P.key1);
        sh.A[k * G + g] = make_float4(x, y, 360.0f * u01(u.x), 0.0f);
    }
}

template <int KIND, int L, int TASK>
__global__ __launch_bounds__(64) void task_step_kernel(const Params P, const Buffers bufs,
                                                       const int n_steps, const int mode) {
    // mode 0: step(action)   1: reset() with random placement   2: open a new episode on the
    // state already in the buffers for the envs flagged in bufs.truncated (reset_to)
    const bool reset_all = mode == 1;
    constexpr int G = 64 / L;
    constexpr int ID = TASK == RSX_TASK_VSS_V0 ? 6 : 8;  // info_dim
    __shared__ Shared<L> sh;
    const int lane = threadIdx.x;
    const int b = lane / G, g = lane % G;
    const int tile = tile_of_block(gridDim.x);
    const int e = tile * G + g;
    const int N = P.n_robots;
    const bool live = e < P.num_envs;
    const bool is_robot = live && b < N, is_ball = live && b == N;
    const size_t B = (size_t)P.num_envs;
    const uint32_t env_id = P.env_id_base + (uint32_t)e;
    const int OD = P.obs_dim;

    // ---- load ----
    Body o; float th_deg;
    load_body<KIND>(P, bufs.state, e, b, is_robot, is_ball, o, th_deg);
    int steps = 0; uint32_t episode = 0;
    if (live) { steps = bufs.steps[e]; episode = bufs.episode[e]; }
    float ou0 = 0.0f, ou1 = 0.0f;
    if (TASK == RSX_TASK_VSS_V0 && is_robot && b >= 1) {
        ou0 = bufs.ou[(size_t)(2 * b) * B + e]; ou1 = bufs.ou[(size_t)(2 * b + 1) * B + e];
    }
    float info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float prev_pot = 0.0f, ep_ret = 0.0f;
    if (is_ball) {
#pragma unroll
        for (int i = 0; i < ID; ++i) info[i] = bufs.info[(size_t)i * B + e];
        prev_pot = bufs.prev_pot[e]; ep_ret = bufs.ep_ret[e];
    }

    float reward = 0.0f; int term = 0, trunc = 0;
    float od = th_deg, wd = 0.0f, wheels[4] = {0, 0, 0, 0};
    bool was_reset = false;
    if (KIND == RSX_KIND_SSL && is_robot) {  // n_steps == 0 or time_step_ms == 0 keep these as loaded
        const float* r = bufs.state + (size_t)(5 + P.rs * b) * B + e;
        wd = r[5 * B];
#pragma unroll
        for (int i = 0; i < 4; ++i) wheels[i] = r[(7 + i) * B];
    } else if (is_robot) {
        wd = bufs.state[(size_t)(5 + P.rs * b + 5) * B + e];
    }

    for (int it = 0; it < n_steps; ++it) {
        bool ended;
        if (mode == 2) {
            const bool flagged = live && bufs.truncated[e] != 0;
            if (flagged) {
                episode += 1; steps = 0; ou0 = 0.0f; ou1 = 0.0f;
                if (is_ball) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) info[i] = 0.0f;
#pragma unroll
                    for (int i = 0; i < ID; ++i) bufs.info[(size_t)i * B + e] = 0.0f;
                    ep_ret = 0.0f; prev_pot = 0.0f;
                }
            }
            write_obs<TASK>(P, sh.stage + g * OD, b, is_robot, is_ball, o.x, o.y, o.vx, o.vy, od, wd, o.ir);
            __syncthreads();
            ended = false;
        } else if (reset_all) {
            // reset(): nothing to simulate; fall through to the placement block below
            ended = live;
            episode += 1;
            if (is_ball) {
#pragma unroll
                for (int i = 0; i < 8; ++i) info[i] = 0.0f;
                ep_ret = 0.0f; prev_pot = 0.0f;
            }
        } else {
            const bool first_step = steps == 0;
            const uint32_t t = (uint32_t)steps;
            if (is_ball && first_step) {
#pragma unroll
                for (int i = 0; i < 8; ++i) info[i] = 0.0f;
                ep_ret = 0.0f;
            }
            const float lastx = o.x, lasty = o.y;  // the reference's last_frame (pre-step)

            // ---- actions -> commands ----
            float q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (TASK == RSX_TASK_VSS_V0) {
                if (is_robot) {
                    float a0, a1;
                    if (b == 0) {
                        if (bufs.actions) { a0 = bufs.actions[(size_t)e * 2]; a1 = bufs.actions[(size_t)e * 2 + 1]; }
                        else {
                            u32x4 u = philox4x32_10(env_id, episode, t, DOM_ACT, P.key0, P.key1);
                            a0 = u01(u.x) * 2.0f - 1.0f; a1 = u01(u.y) * 2.0f - 1.0f;
                        }
                    } else {  // Ornstein-Uhlenbeck noise, Utils/Utils.py:14-21 (Box-Muller on Philox)
                        u32x4 u = philox4x32_10(env_id, episode, t, DOM_OU | ((uint32_t)b << 8), P.key0, P.key1);
                        float u1 = (float)((u.x >> 8) + 1u) * 5.9604644775390625e-08f;
                        float ang = (u01(u.y) - 0.5f) * P.two_pi;
                        float rad = sqrtf(-2.0f * log_f32(u1));
                        float sn, cs;
                        sincos_f32(ang, sn, cs);
                        float n0 = rad * cs, n1 = rad * sn;
                        ou0 = (ou0 + P.ou_theta_dt * (0.0f - ou0)) + P.ou_sig_sqdt * n0;
                        ou1 = (ou1 + P.ou_theta_dt * (0.0f - ou1)) + P.ou_sig_sqdt * n1;
                        a0 = ou0; a1 = ou1;
                    }
                    q[0] = vss_wheel(P, a0); q[1] = vss_wheel(P, a1);
                }
            } else {  // static_defenders.py:114-148
                if (is_robot && b == 0) {
                    float a[5];
                    if (bufs.actions) { for (int i = 0; i < 5; ++i) a[i] = bufs.actions[(size_t)e * 5 + i]; }
                    else {
                        u32x4 u = philox4x32_10(env_id, episode, t, DOM_ACT, P.key0, P.key1);
                        a[0] = u01(u.x) * 2.0f - 1.0f; a[1] = u01(u.y) * 2.0f - 1.0f;
                        a[2] = u01(u.z) * 2.0f - 1.0f; a[3] = u01(u.w) * 2.0f - 1.0f;
                        u32x4 v = philox4x32_10(env_id, episode, t, DOM_ACT | (1u << 8), P.key0, P.key1);
                        a[4] = u01(v.x) * 2.0f - 1.0f;
                    }
                    float sn, cs;
                    sincos_f32(od * P.deg2rad, sn, cs);
                    float gx = a[0] * P.max_v, gy = a[1] * P.max_v, vth = a[2] * 10.0f;
                    float lx = gx * cs + gy * sn, ly = gy * cs - gx * sn;
                    float nrm = sqrtf(lx * lx + ly * ly);
                    if (!(nrm < P.max_v)) { float sc = P.max_v / nrm; lx = lx * sc; ly = ly * sc; }
                    q[1] = lx; q[2] = ly; q[3] = vth;
                    q[5] = a[3] > 0.0f ? 5.0f : 0.0f;
                    q[7] = a[4] > 0.0f ? 1.0f : 0.0f;
                }
            }
            if (is_robot) robot_targets<KIND>(P, o, q);

            // ---- physics ----
            physics<KIND, L>(P, o, b, g, live, sh);

            // ---- wire-format values, observation, reward ----
            if (is_robot) {
                od = o.th * P.rad2deg; wd = o.om * P.rad2deg;
                if (KIND == RSX_KIND_SSL) {
                    float vf = o.vx * o.c + o.vy * o.s;
                    float vl = o.vy * o.c - o.vx * o.s;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        wheels[i] = ((vl * P.wc[i] - vf * P.ws[i]) + o.om * P.r_robot) * P.inv_rw;
                    // a round trip through the wire format: theta is kept in degrees in HBM
                }
                // keep the lane's internal angle consistent with what a reload would give
                o.th = od * P.deg2rad; o.om = wd * P.deg2rad;
                sincos_f32(o.th, o.s, o.c);
            } else if (is_ball) {
                // height goes through the wire format too: z_wire = r_ball + z
                o.z = (P.r_ball + o.z) - P.r_ball;
            }
            write_obs<TASK>(P, sh.stage + g * OD, b, is_robot, is_ball, o.x, o.y, o.vx, o.vy, od, wd, o.ir);
            if (is_robot && b == 0) {
                float* xr = sh.x0[g];
                xr[0] = o.x; xr[1] = o.y; xr[2] = o.vx; xr[3] = o.vy; xr[4] = q[0]; xr[5] = q[1];
                xr[6] = lastx; xr[7] = lasty;
                xr[8] = wheels[0]; xr[9] = wheels[1]; xr[10] = wheels[2]; xr[11] = wheels[3];
            }
            __syncthreads();
            if (is_ball) {
                const float* xr = sh.x0[g];
                reward = 0.0f; term = 0;
                const float bx = o.x, by = o.y;
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
                        float mv = (rbx / nrm) * xr[2] + (rby / nrm) * xr[3];
                        float move = clampf(mv * 2.5f, -5.0f, 5.0f);
                        float energy = -(fabsf(xr[4]) + fabsf(xr[5]));
                        float t_move = 0.2f * move, t_grad = 0.8f * grad, t_en = 2e-4f * energy;
                        reward = (t_move + t_grad) + t_en;
                        info[1] += t_move; info[2] += t_grad; info[3] += t_en;
                    }
                } else {  // static_defenders.py:150-212,256-322
                    const float rx = xr[0], ry = xr[1];
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
                        float en = -(((fabsf(xr[8]) + fabsf(xr[9])) + fabsf(xr[10])) + fabsf(xr[11])) * P.inv_en_scale;
                        info[5] += bd; info[6] += bg; info[7] += en;
                        reward = (bd + bg) + en;
                    }
                }
                ep_ret = ep_ret + reward;
            }
            steps += 1;
            trunc = steps >= P.max_steps;
            if (is_ball) sh.flag[g] = term | trunc;
            __syncthreads();
            ended = live && sh.flag[g] != 0;
            if (is_ball) {
                // info is reported as it stands after this step (cleared lazily at the next
                // episode's first step), like the dict the reference returns with `done`
#pragma unroll
                for (int i = 0; i < ID; ++i) bufs.info[(size_t)i * B + e] = info[i];
                bufs.reward[e] = reward; bufs.terminated[e] = (uint8_t)term; bufs.truncated[e] = (uint8_t)trunc;
            }
        }

        // ---- episode end: same-step auto-reset (or reset()) ----
        if (__any(ended)) {
            if (ended && !reset_all) {  // terminal observation
                for (int i = b; i < OD; i += L) bufs.final_obs[(size_t)e * OD + i] = sh.stage[g * OD + i];
            }
            if (ended && is_ball) {
                if (!reset_all) {
                    episode += 1;
                    atomicAdd(&bufs.metrics[1], 1ull);
                    if (TASK == RSX_TASK_VSS_V0) {
                        if (info[4] > 0.0f) atomicAdd(&bufs.metrics[2], 1ull);
                        if (info[5] > 0.0f) atomicAdd(&bufs.metrics[3], 1ull);
                    } else if (info[0] > 0.0f) atomicAdd(&bufs.metrics[2], 1ull);
                    atomicAdd(&bufs.metrics[4], (unsigned long long)__float2ll_rn(ep_ret * 1048576.0f));
                    atomicAdd(&bufs.metrics[5], (unsigned long long)steps);
                    if (trunc && !term) atomicAdd(&bufs.metrics[6], 1ull);
                }
            }
            __syncthreads();  // stage rows of ended envs are about to be overwritten
            if (ended && is_ball) place_env<TASK, L>(P, env_id, episode, g, sh);
            __syncthreads();
            if (ended) {
                if (!is_ball && !reset_all) episode += 1;
                steps = 0; ou0 = 0.0f; ou1 = 0.0f; was_reset = true;
                if (is_robot || is_ball) {
                    const float4 pz = sh.A[b * G + g];
                    o = Body{};
                    o.x = pz.x; o.y = pz.y;
                    od = pz.z; wd = 0.0f;
                    wheels[0] = wheels[1] = wheels[2] = wheels[3] = 0.0f;
                    if (is_robot) { o.th = od * P.deg2rad; sincos_f32(o.th, o.s, o.c); }
                }
                write_obs<TASK>(P, sh.stage + g * OD, b, is_robot, is_ball, o.x, o.y, o.vx, o.vy, od, wd, 0);
            }
            __syncthreads();
        }

        // ---- observation out, coalesced: the tile's G rows are one contiguous run ----
        {
            const size_t base = (size_t)tile * G * OD;
            const size_t lim = B * (size_t)OD;
            for (int i = lane; i < G * OD; i += 64)
                if (base + i < lim) bufs.obs[base + i] = sh.stage[i];
        }
        __syncthreads();
    }

    // ---- store (wire format: degrees, deg/s; SSL: infrared + wheel speeds) ----
    if (mode == 2) {
        // state untouched
    } else if (is_robot) {
        float* r = bufs.state + (size_t)(5 + P.rs * b) * B + e;
        r[0] = o.x; r[B] = o.y; r[2 * B] = od; r[3 * B] = o.vx; r[4 * B] = o.vy; r[5 * B] = wd;
        if (KIND == RSX_KIND_SSL) {
            if (P.n_sub || was_reset) r[6 * B] = o.ir ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[(7 + i) * B] = wheels[i];
        }
    } else if (is_ball) {
        float* r = bufs.state + e;
        r[0] = o.x; r[B] = o.y; r[2 * B] = P.r_ball + o.z; r[3 * B] = o.vx; r[4 * B] = o.vy;
        r[(size_t)P.state_dim * B] = o.vz;
    }
    if (live && b == 0) { bufs.steps[e] = steps; bufs.episode[e] = episode; }
    if (TASK == RSX_TASK_VSS_V0 && is_robot && b >= 1) {
        bufs.ou[(size_t)(2 * b) * B + e] = ou0; bufs.ou[(size_t)(2 * b + 1) * B + e] = ou1;
    }
    if (is_ball) { bufs.prev_pot[e] = prev_pot; bufs.ep_ret[e] = ep_ret; }
}

}  // namespace rsx
