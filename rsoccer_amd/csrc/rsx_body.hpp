// rsx_body.hpp — the per-body arithmetic of the step model, stated ONCE for the device.
//
// Everything a kernel does to one rigid body during a sub-step lives here and nowhere else: actuation and
// integration of a robot (differential drive / holonomic), the ball's flight and its once-per-step rolling
// resistance, the response of a body to one touching partner (normal impulse, Coulomb friction, de-penetration),
// the robot - ball contact of the SSL robots (kicker mouth, infrared, kick / dribbler), the wall clamp and the
// ball's wall friction.  The three kernel families — lane groups (rsx_kernels.hpp), one lane per env
// (rsx_epl.hpp, rsx_epl_ssl.hpp) and four lanes per env (rsx_quad_ssl.hpp) — differ in WHERE a body lives and how
// partners are found; they all call these functions on a `Body`, so a change of the model is one edit, and the
// layouts cannot drift apart (they are compared bit for bit in tests/test_gpu_parity.py anyway).
// (The executable DEFINITION of the model is a separate CPU restatement under the test infrastructure, which shares
// no code with this file: DESIGN.md section 2.)  Reference call sites replaced: robosim.VSS.step / robosim.SSL.step, rsoccer_gym/Simulators/
// rsim.py:102,155.
#pragma once
#include <hip/hip_runtime.h>

#include "rsx_math.hpp"
#include "rsx_params.hpp"

namespace rsx {

// what one lane keeps in registers for its body
struct Body {
    float x, y, vx, vy;       // all bodies
    float th, om, c, s;       // robots: heading (DEGREES, the wire unit), rate (rad/s), cos/sin(heading)
    float t0, t1, t2;         // VSS: v target, omega target | SSL: local vx, vy, omega targets
    float kick_x, kick_z;     // SSL
    float z, vz;              // ball: height above rest, vertical speed
    int drib, ir;
};


// clamp a circle (radius r, restitution rest) into the playable region
template <int KIND>
__device__ __forceinline__ void walls(const Params& P, const float r, const float rest, float& x,
                                      float& y, float& vx, float& vy, int& hit /* bit 0: vx reflected, bit 1: vy */) {
    using K = KC<KIND>;
    float ax = fabsf(x), ay = fabsf(y);
    const float sx = signf(x), sy = signf(y);  // only read when |x| (|y|) exceeds a positive limit
    if (KIND == RSX_KIND_VSS) {
        // predicated form of: inside a goal box (|x| > L/2) the limits are the goal's side and
        // back walls, otherwise the touch line and - outside the goal mouth - the goal line
        const bool in_goal = ax > P.half_len;
        const float yl = (in_goal ? P.ghw : P.half_wid) - r;
        const float xl = (in_goal ? P.half_len + P.gd : P.half_len) - r;
        const bool hy = ay > yl;
        const bool hx = (ax > xl) & (in_goal | (ay > P.ghw - r));
        const float ny = sy * yl, nx = sx * xl;
        const bool fy = hy & (vy * sy > 0.0f), fx = hx & (vx * sx > 0.0f);
        y = hy ? ny : y; vy = fy ? -rest * vy : vy;
        x = hx ? nx : x; vx = fx ? -rest * vx : vx;
        hit = (fx ? 1 : 0) | (fy ? 2 : 0);
    } else {
        // predicated like the VSS clamp: conditional stores to x / y / vx / vy inside nested
        // branches get merged by the compiler into stores through a selected POINTER, which
        // pins the body's velocity in scratch memory (a global-memory round trip per access)
        const float yl = (P.half_wid + K::margin) - r, xl = (P.half_len + K::margin) - r;
        const bool hy = ay > yl;
        const bool fy0 = hy & (vy * sy > 0.0f);
        y = hy ? sy * yl : y; vy = fy0 ? -rest * vy : vy; ay = hy ? yl : ay;
        const bool hx = ax > xl;
        const bool fx0 = hx & (vx * sx > 0.0f);
        x = hx ? sx * xl : x; vx = fx0 ? -rest * vx : vx; ax = hx ? xl : ax;
        hit = (fx0 ? 1 : 0) | (fy0 ? 2 : 0);
        if (ax > P.half_len) {   // beyond a goal line: the goal's walls (rare)
            const float back = P.half_len + P.gd;
            const bool in_mouth = ay < P.ghw;
            const bool inside = in_mouth & (ax < back);
            const bool c1 = inside & (ax > back - r);                       // back wall, from inside
            const bool c2 = inside & (ay > P.ghw - r);                      // side wall, from inside
            const bool c3 = in_mouth & !(ax < back) & (ax < back + r);      // behind the back wall
            const bool c4 = !in_mouth & (ay < P.ghw + r) & (ax < back);     // outside, touching a side wall
            const float vxs = vx * sx, vys = vy * sy;
            const bool fx = (c1 & (vxs > 0.0f)) | (c3 & (vxs < 0.0f));
            const bool fy = (c2 & (vys > 0.0f)) | (c4 & (vys < 0.0f));
            x = c1 ? sx * (back - r) : (c3 ? sx * (back + r) : x);
            y = c2 ? sy * (P.ghw - r) : (c4 ? sy * (P.ghw + r) : y);
            vx = fx ? -rest * vx : vx;
            vy = fy ? -rest * vy : vy;
            hit |= (fx ? 1 : 0) | (fy ? 2 : 0);
        }
    }
}

// A bounce of the BALL off a wall with Coulomb friction at the contact point: couples the velocity
// component along the wall with the spin about the vertical axis.  (vx0, vy0) = velocity before
// walls(): the ball moved INTO the wall, so its sign names the wall's side.
template <int KIND>
__device__ __forceinline__ void ball_wall_spin(const int hit, const float vx0, const float vy0,
                                               float& vx, float& vy, float& om) {
    using K = KC<KIND>;
    if (hit & 2) {
        const float sg = vy0 < 0.0f ? -1.0f : 1.0f;
        const float vc = vx - (om * K::r_ball) * sg;
        const float lim = K::mu_wb * (K::ope_wb * fabsf(vy0));
        const float d = clampf(-(vc * K::kw), -lim, lim);
        vx = vx + d; om = om - (sg * d) * K::spin_c;
    }
    if (hit & 1) {
        const float sg = vx0 < 0.0f ? -1.0f : 1.0f;
        const float vc = vy + (om * K::r_ball) * sg;
        const float lim = K::mu_wb * (K::ope_wb * fabsf(vx0));
        const float d = clampf(-(vc * K::kw), -lim, lim);
        vy = vy + d; om = om + (sg * d) * K::spin_c;
    }
}

// per-step command processing of a robot lane: wheel / velocity commands -> targets
template <int KIND>
__device__ __forceinline__ void robot_targets(const Params& P, Body& o, const float* q /*C cmds*/) {
    using K = KC<KIND>;
    if (KIND == RSX_KIND_VSS) {
        float wl = clampf(q[0], -K::w_max, K::w_max);
        float wr = clampf(q[1], -K::w_max, K::w_max);
        o.t0 = (wl + wr) * K::half_rw;
        o.t1 = (wr - wl) * K::rw_2b;
        o.t2 = 0.0f; o.kick_x = 0.0f; o.kick_z = 0.0f; o.drib = 0;
    } else {
        float vtx, vty, omt;
        if (q[0] != 0.0f) {
            float w0 = clampf(q[1], -K::w_max, K::w_max), w1 = clampf(q[2], -K::w_max, K::w_max);
            float w2 = clampf(q[3], -K::w_max, K::w_max), w3 = clampf(q[4], -K::w_max, K::w_max);
            vtx = (((P.pinv[0][0] * w0 + P.pinv[0][1] * w1) + P.pinv[0][2] * w2) + P.pinv[0][3] * w3) * K::r_wheel;
            vty = (((P.pinv[1][0] * w0 + P.pinv[1][1] * w1) + P.pinv[1][2] * w2) + P.pinv[1][3] * w3) * K::r_wheel;
            omt = (((P.pinv[2][0] * w0 + P.pinv[2][1] * w1) + P.pinv[2][2] * w2) + P.pinv[2][3] * w3) * K::r_wheel;
        } else {
            vtx = q[1]; vty = q[2]; omt = q[3];
            float m = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float wi = ((vty * P.wc[i] - vtx * P.ws[i]) + omt * K::r_robot) * K::inv_rw;
                float a = fabsf(wi);
                if (a > m) m = a;
            }
            if (m > K::w_max) { float sc = K::w_max / m; vtx = vtx * sc; vty = vty * sc; omt = omt * sc; }
        }
        o.t0 = vtx; o.t1 = vty; o.t2 = omt;
        o.kick_x = q[5]; o.kick_z = q[6]; o.drib = q[7] != 0.0f;
    }
}

// Response of a body to ONE touching partner, from the body's point of view (each side of a pair
// evaluates this with its own constants).  n = unit normal body -> partner, pen = penetration,
// (dvx, dvy) = v_partner - v_body, wsum = om_partner * lever_partner + om_body * lever_body (surface
// speeds at the contact point), w / kt = the body's share of the normal / tangential impulse,
// mu = Coulomb coefficient, spin_c = spin per unit of tangential velocity change (ball only).
__device__ __forceinline__ void respond(const float nx, const float ny, const float pen, const float dvx,
                                        const float dvy, const float wsum, const float ope, const float w,
                                        const float kt, const float mu, const float spin_c, const float beta,
                                        float& avx, float& avy, float& apx, float& apy, float& aw) {
    float vn = fma_(dvx, nx, dvy * ny);
    if (vn < 0.0f) {
        float q = ope * vn * w;                           // <= 0: pushes the body away from the partner
        avx = fma_(q, nx, avx); avy = fma_(q, ny, avy);
        float vt = fma_(dvy, nx, -(dvx * ny)) - wsum;     // along t = (-ny, nx)
        float lim = q * mu;
        float ft = clampf(vt * kt, lim, -lim);            // sticking impulse, Coulomb-limited
        avx = fma_(-ft, ny, avx); avy = fma_(ft, nx, avy);
        aw = fma_(ft, spin_c, aw);
    }
    float pc = beta * pen * w;
    apx = fma_(-pc, nx, apx); apy = fma_(-pc, ny, apy);
}

// circle - circle pair known to overlap (d2 = squared centre distance)
__device__ __forceinline__ void contact_response(const Body& o, const float4 oj, const float d2,
                                                 const float rs, const float ope, const float w,
                                                 const float kt, const float mu, const float spin_c,
                                                 const float wsum, const float beta, const float pen2,
                                                 float& avx, float& avy, float& apx, float& apy, float& aw,
                                                 bool& deep) {
    float dx = oj.x - o.x, dy = oj.y - o.y;
    float d = sqrtf(d2), inv = 1.0f / d;
    respond(dx * inv, dy * inv, rs - d, oj.z - o.vx, oj.w - o.vy, wsum, ope, w, kt, mu, spin_c, beta,
            avx, avy, apx, apy, aw);
    deep |= rs - d > pen2;   // an impact at speed or a jammed pile: the env gets a second sweep
}


// Both sides of ONE touching pair (i, j) in one go — for the layouts in which a lane holds both bodies.  Each side's
// result is bit for bit what contact_response computes from that body's point of view: the geometry of the two
// views differs only by sign (x_i - x_j = -(x_j - x_i) exactly, so d2, d, 1/d, the penetration, the normal speed vn
// and the tangential term fma(dvy, nx, -(dvx ny)) are the same numbers; the normals are each other's negation),
// and a negated operand is free.  Shared: one square root, one division, one normal; per side: the shares w / kt,
// the Coulomb limit, the surface-speed term wsum (whose rounding depends on the side) and the sums.
// ai = {vx, vy, px, py} sums of body i (a robot: no spin), aj the same of body j, awj the spin sum of j (ball only).
__device__ __forceinline__ void contact_pair(const Body& bi, const Body& bj, const float wsum_i, const float wsum_j,
                                             const float rs, const float ope, const float w_i, const float w_j,
                                             const float kt_i, const float kt_j, const float mu, const float spin_c_j,
                                             const float beta, const float pen2, float* ai, float* aj, float& awj, bool& deep) {
    const float dx = bj.x - bi.x, dy = bj.y - bi.y;
    const float d = sqrtf(fma_(dx, dx, dy * dy)), inv = 1.0f / d;
    const float nx = dx * inv, ny = dy * inv, pen = rs - d;           // n: i -> j
    const float dvx = bj.vx - bi.vx, dvy = bj.vy - bi.vy;
    const float vn = fma_(dvx, nx, dvy * ny);
    if (vn < 0.0f) {
        const float vt0 = fma_(dvy, nx, -(dvx * ny));
        {   // body i
            const float q = ope * vn * w_i;
            ai[0] = fma_(q, nx, ai[0]); ai[1] = fma_(q, ny, ai[1]);
            const float lim = q * mu;
            const float ft = clampf((vt0 - wsum_i) * kt_i, lim, -lim);
            ai[0] = fma_(-ft, ny, ai[0]); ai[1] = fma_(ft, nx, ai[1]);
        }
        {   // body j: normal -n
            const float q = ope * vn * w_j;
            aj[0] = fma_(q, -nx, aj[0]); aj[1] = fma_(q, -ny, aj[1]);
            const float lim = q * mu;
            const float ft = clampf((vt0 - wsum_j) * kt_j, lim, -lim);
            aj[0] = fma_(ft, ny, aj[0]); aj[1] = fma_(ft, -nx, aj[1]);
            awj = fma_(ft, spin_c_j, awj);
        }
    }
    const float pci = beta * pen * w_i, pcj = beta * pen * w_j;
    ai[2] = fma_(-pci, nx, ai[2]); ai[3] = fma_(-pci, ny, ai[3]);
    aj[2] = fma_(pcj, nx, aj[2]); aj[3] = fma_(pcj, ny, aj[3]);
    deep |= pen > pen2;   // an impact at speed or a jammed pile: the env gets a second sweep
}

// ---- one sub-step of a robot: actuation towards its targets + integration of velocity and heading ----
// (position: advance_body).  VSS (differential drive): forward speed towards t0 with an acceleration cap, lateral
// speed removed with a grip cap, yaw rate towards t1.  SSL (holonomic): robot-frame velocity towards (t0, t1) with
// a vector acceleration cap, yaw rate towards t2.  The heading is integrated in degrees (the wire unit); its
// cosine / sine are carried by a small rotation (one exact sincos per step()).
template <int KIND>
__device__ __forceinline__ void actuate_robot(const Params& P, Body& o) {
    float vf = fma_(o.vy, o.s, o.vx * o.c);
    float vl = fma_(o.vy, o.c, -(o.vx * o.s));
    if (KIND == RSX_KIND_VSS) {
        vf = vf + clampf(o.t0 - vf, -P.a_lin_h, P.a_lin_h);
        vl = vl - clampf(vl, -P.a_lat_h, P.a_lat_h);
        o.om = o.om + clampf(o.t1 - o.om, -P.a_ang_h, P.a_ang_h);
    } else {
        float dx = o.t0 - vf, dy = o.t1 - vl;
        float d2 = fma_(dx, dx, dy * dy);
        if (d2 > P.a_lin_h2) { float sc = P.a_lin_h / sqrtf(d2); dx = dx * sc; dy = dy * sc; }
        vf = vf + dx; vl = vl + dy;
        o.om = o.om + clampf(o.t2 - o.om, -P.a_ang_h, P.a_ang_h);
    }
    o.vx = fma_(vf, o.c, -(vl * o.s));
    o.vy = fma_(vf, o.s, vl * o.c);
}
// heading of a robot after the sub-step's turn (degrees, wrapped) — separate from actuate_robot because a kernel
// may keep the heading somewhere else than the rest of the body (rsx_epl.hpp parks it in LDS)
__device__ __forceinline__ float advance_heading(const Params& P, const float om, const float th) {
    return wrap_deg(fma_(om, P.h_deg, th));
}
template <int KIND>
__device__ __forceinline__ void integrate_robot(const Params& P, Body& o) {   // the whole robot part of phase A
    actuate_robot<KIND>(P, o);
    o.x = fma_(o.vx, P.h, o.x);
    o.y = fma_(o.vy, P.h, o.y);
    o.th = advance_heading(P, o.om, o.th);
    rotate_heading(o.om * P.h, o.c, o.s);
}
// the ball's phase A: ballistic flight with bounces while it is off the ground, then the position advance
__device__ __forceinline__ void ball_flight(const Params& P, Body& ball, const float e_ground, const float vz_min) {
    ball.vz = ball.vz - P.g_h;
    ball.z = fma_(ball.vz, P.h, ball.z);
    if (ball.z <= 0.0f) {
        ball.z = 0.0f;
        ball.vz = -ball.vz * e_ground;
        if (ball.vz < vz_min) ball.vz = 0.0f;
    }
}
template <int KIND>
__device__ __forceinline__ void integrate_ball(const Params& P, Body& ball) {
    using K = KC<KIND>;
    if (ball.z > 0.0f || ball.vz > 0.0f) ball_flight(P, ball, K::e_ground, K::vz_min);
    ball.x = fma_(ball.vx, P.h, ball.x);
    ball.y = fma_(ball.vy, P.h, ball.y);
}
// once per step(), before the sub-steps: rolling resistance (a constant deceleration to an exact stop) while the
// ball is on the ground, and the decay of its spin about the vertical axis
__device__ __forceinline__ void ball_step_friction(const Params& P, Body& ball) {
    if (P.n_sub && !(ball.z > 0.0f || ball.vz > 0.0f)) {
        float sp2 = fma_(ball.vx, ball.vx, ball.vy * ball.vy);
        if (sp2 > 0.0f) {
            float sp = sqrtf(sp2), ns = sp - P.mu_g_dt;
            if (ns < 0.0f) ns = 0.0f;
            float kk = ns / sp;
            ball.vx = ball.vx * kk; ball.vy = ball.vy * kk;
        }
        const float aw = fabsf(ball.om) - P.spin_dec_dt;
        ball.om = aw > 0.0f ? (ball.om < 0.0f ? -aw : aw) : 0.0f;
    }
}
// SSL: is this body anywhere near a wall?  walls<SSL> changes nothing while |x| <= half_len (no goal geometry) and |y| <=
// half_wid + margin - r_robot (the tightest of the y limits, robots' and ball's): kernels test a whole wave's bodies with
// this and skip the clamp when none is out there (the division-A field is 12 m x 9 m: most sub-steps).  NaN ("ghost"
// slots of rsx_quad_ssl.hpp) compares false.
template <int KIND>
__device__ __forceinline__ bool near_walls(const Params& P, const float x, const float y) {
    using K = KC<KIND>;
#ifdef RSX_NO_WALL_SKIP   // development A/B: always run the clamp
    return true;
#endif
    return (fabsf(x) > P.half_len) | (fabsf(y) > (P.half_wid + K::margin) - K::r_robot);
}
// the ball's wall clamp incl. the friction of a bounce
template <int KIND>
__device__ __forceinline__ void ball_walls(const Params& P, Body& ball) {
    using K = KC<KIND>;
    const float vx0 = ball.vx, vy0 = ball.vy;
    int hit = 0;
    walls<KIND>(P, K::r_ball, K::e_wb, ball.x, ball.y, ball.vx, ball.vy, hit);
    if (hit) ball_wall_spin<KIND>(hit, vx0, vy0, ball.vx, ball.vy, ball.om);
}
template <int KIND>
__device__ __forceinline__ void robot_walls(const Params& P, Body& o) {
    using K = KC<KIND>;
    int hit;
    walls<KIND>(P, K::r_robot, K::e_wr, o.x, o.y, o.vx, o.vy, hit);
}

}  // namespace rsx
