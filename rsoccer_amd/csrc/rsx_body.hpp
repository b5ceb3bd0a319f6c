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
// VSS goal post (model v2), the response: a body inside the chord of a post's arc (|x| + |y| > (L/2 + goal half width) - r in the corner
// region: walls<VSS> below has the gate) is pushed out along the chord's normal (1, 1) / sqrt 2, folded into the first quadrant, and the
// velocity component along it is reflected.  Called under the gate (per lane).
__device__ __forceinline__ void post_response(const Params& P, const float r, const float rest, float& x, float& y, float& vx, float& vy, int& hit) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float sx = signf(x), sy = signf(y);
    const float hf = 0.5f * ((ax + ay) - ((P.half_len + P.ghw) - r));
    x = sx * (ax - hf); y = sy * (ay - hf);
    const float vn = fma_(vx, sx, vy * sy);   // sqrt 2 x the speed along the normal; > 0: moving into the post
    if (vn > 0.0f) {
        const float dv = ((1.0f + rest) * 0.5f) * vn;
        vx = fma_(-sx, dv, vx); vy = fma_(-sy, dv, vy);
        hit |= 4;
    }
}

// DEFER_POST (VSS): the goal-post response is left to the caller (post_response below), who merges its rare branch with one it has
// anyway; bit 3 of `hit` then says "this body is inside a post's chord".
template <int KIND, bool DEFER_POST = false>
__device__ __forceinline__ void walls(const Params& P, const float r, const float rest, float& x,
                                      float& y, float& vx, float& vy, int& hit /* bit 0: vx reflected, bit 1: vy, bit 2: off a post */) {
    using K = KC<KIND>;
    float ax = fabsf(x), ay = fabsf(y);
    const float sx = signf(x), sy = signf(y);  // only read when |x| (|y|) exceeds a positive limit
    if (KIND == RSX_KIND_VSS) {
        // predicated form of: inside a goal box (|x| > L/2) the limits are the goal's side and
        // back walls, otherwise the touch line and - beside the goal mouth (|y| >= goal half width) - the goal line
        const bool in_goal = ax > P.half_len;
        const float yl = (in_goal ? P.ghw : P.half_wid) - r;
        const float xl = (in_goal ? P.half_len + P.gd : P.half_len) - r;
        const bool hy = ay > yl;
        const bool faced = in_goal | (ay >= P.ghw);   // a wall faces the body on the x axis: the goal's back wall, or the goal line's beside the mouth
        const bool hx = (ax > xl) & faced;
        const bool fy = hy & (vy * sy > 0.0f), fx = hx & (vx * sx > 0.0f);
        y = hy ? sy * yl : y; vy = fy ? -rest * vy : vy;
        x = hx ? sx * xl : x; vx = fx ? -rest * vx : vx;
        hit = (fx ? 1 : 0) | (fy ? 2 : 0);
        // goal posts (model v2): the goal line's wall ends at the mouth in a corner (+-L/2, +-goal half width) that a body keeps its radius
        // from — along the CHORD of that arc: with u, v the centre's distances to the goal line and to the mouth's edge, u + v >= r, i.e.
        // |x| + |y| <= (L/2 + goal half width) - r.  The chord joins the goal line's limit to the goal's side-wall limit, so the limits
        // are continuous all the way round (a body cuts the corner by at most 0.29 r); no square root, no division.  (v1 clamped x for
        // |y| > goal half width - r: a body sliding along the mouth's edge into that strip was thrown up to a radius sideways into whatever
        // stood there — the 3 cm overlaps of the VSS scrum.)  Robots rest against posts a lot (the scrum of a goal mouth) and a lone wave
        // pays ~6 cycles for every instruction it issues: the gate is one add and one compare next to the clamp's own, the response is
        // short and behind a wave-uniform branch.  Pushed out along the chord's normal (1, 1) / sqrt 2, folded into the first quadrant.
        const float cl = (P.half_len + P.ghw) - r;
        const float sa = ax + ay;
        const bool inp = !faced & (sa > cl);
        if (DEFER_POST) { hit |= inp ? 8 : 0; }
        else if (__builtin_expect(__any(inp), 0)) {
            if (inp) post_response(P, r, rest, x, y, vx, vy, hit);
        }
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
        {   // goal posts (model v2): the open ends of the goal's side walls are points at (+-L/2, +-goal_width/2) that a body keeps
            // its radius from — without them a body overlaps the wall's end from the field side and is thrown sideways by the
            // side-wall clamp the moment it crosses the goal line.  Folded into the first quadrant, n points post -> body;
            // predicated like the rest of the clamp (a garbage sqrt / divide of a far-away body is selected away).
            const float dxp = ax - P.half_len, dyp = ay - P.ghw;
            const float d2 = fma_(dxp, dxp, dyp * dyp);
            const bool inp = (d2 < r * r) & (d2 > 0.0f);
            if (__builtin_expect(__any(inp), 0)) {   // wave-uniform: the square root and the division stay off the clamp's usual path
            const float d = sqrtf(d2), inv = 1.0f / d;
            const float nxp = dxp * inv, nyp = dyp * inv;
            ax = inp ? fma_(r, nxp, P.half_len) : ax; ay = inp ? fma_(r, nyp, P.ghw) : ay;
            x = inp ? sx * ax : x; y = inp ? sy * ay : y;
            const float vr = fma_(vx * sx, nxp, (vy * sy) * nyp);   // radial speed; < 0: moving into the post
            const bool fp = inp & (vr < 0.0f);
            const float dv = -((1.0f + rest) * vr);
            vx = fp ? fma_(sx, dv * nxp, vx) : vx; vy = fp ? fma_(sy, dv * nyp, vy) : vy;
            hit |= fp ? 4 : 0;
            }
        }
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

// Is this body anywhere near a wall?  walls<KIND> with a robot's radius (and, a fortiori, the ball's) changes nothing outside this
// region — |x| <= L/2 - r_robot (SSL: no goal post within reach, no goal geometry) and |y| <= W/2 + margin - r_robot — so kernels test a whole wave's bodies with this and skip the clamp when none is out there (the division-A
// field is 12 m x 9 m: most sub-steps).  NaN ("ghost" slots of rsx_quad_ssl.hpp) compares false.
template <int KIND>
__device__ __forceinline__ bool near_walls(const Params& P, const float x, const float y) {
    using K = KC<KIND>;
    // one form for both classes (K::margin is 0 for VSS).  SSL: the strip of a robot's radius before the goal lines is there for the goal
    // posts only; testing |y| against the posts as well would be exact, but costs two more DEPENDENT instructions at the end of every
    // sub-step of every wave (measured: +1.3 % on the 1v6 and 11v11 steps at the latency-bound batches) to save a rare clamp
    return (fabsf(x) > P.half_len - K::r_robot) | (fabsf(y) > (P.half_wid + K::margin) - K::r_robot);
}

// The same gate without the loose strip — SSL: beyond a goal line, at a boundary wall, or within a robot's radius of a goal post.  For
// the large-batch kernels of the many-robot tasks (one / four lanes per env), which are bound by instruction issue, not by the length of
// a wave's dependent chain: there the extra instructions per body are cheaper than the eight (or twenty-three) clamps a wave runs when
// some body sits in the strip before a goal line (SSLStaticDefenders: the ball on its way into the goal).
template <int KIND>
__device__ __forceinline__ bool near_walls_exact(const Params& P, const float x, const float y) {
    using K = KC<KIND>;
    if (KIND == RSX_KIND_VSS) return near_walls<KIND>(P, x, y);
    return (fabsf(x) > P.half_len) | (fabsf(y) > (P.half_wid + K::margin) - K::r_robot) |
           ((fabsf(x) > P.half_len - K::r_robot) & (fabsf(fabsf(y) - P.ghw) < K::r_robot));
}

// Is a robot's centre within 2 mm of where the wall clamp acts on it (or beyond)?  The probes of wall_shares lie within 1 mm of the
// body, so a pair of which NEITHER body is at_wall has unblocked probes by construction and keeps the plain 1/2 : 1/2 shares —
// the limits are those of the clamp minus 2 mm (one of them rounding slack).  (Computed from the fields the kernels hold anyway: two
// more words in the parameter block cost the SSL task kernels scalar registers they do not have.)
template <int KIND>
__device__ __forceinline__ bool at_wall(const Params& P, const float x, const float y) {
    using K = KC<KIND>;
    return (fabsf(x) > P.half_len - (K::r_robot + 0.002f)) | (fabsf(y) > P.half_wid + (K::margin - K::r_robot - 0.002f));
}

// Model v2: a robot that stands against a wall cannot yield along that wall's normal — in a robot - robot pair the partner then takes
// the whole correction on that axis (the wall holds the other side).  Without this a pile that the robots' own push presses against a
// wall overlaps by centimetres: the contact phase moves the outer robot into the wall, the wall clamp puts it back into its neighbour.
// Blocked axes are found by PROBING: each body displaced by 1 mm the way this contact pushes it (a: against n, p: along n); an axis
// is blocked when the probe lies where the wall clamp acts on that coordinate (blocked_axes below).  Shares per axis: blocked body 0, its free partner 1,
// otherwise 1/2 each.  Returns whether any axis of either body was blocked (a "wall pair": such envs may take a third and a fourth
// sweep).  Callers evaluate it only for pairs with a body at_wall (everywhere else the result is 1/2, 1/2, false).
// does the wall clamp (walls<SSL>, a robot's radius) act on the x / the y coordinate of this point?
//   x: beyond the end wall's limit | within r of a goal post | in a goal: at its back wall from inside, or behind it
//   y: beyond the side wall's limit | within r of a goal post | in a goal: at a side wall from inside, or beside it from outside
// Every predicate "a > b" is the SIGN BIT of the float b - a (exact: a difference of finite floats is negative iff a > b), combined
// with integer AND / OR / NOT in vector registers; the answer is bit 31 of bx / by.  Written as compares, each truth value is a
// 64-bit lane mask in a scalar-register pair: two probes x fifteen predicates live at once pushed the lane-group kernels out of
// scalar registers (176 v_writelane / v_readlane in the 32-lane sub-step loop, +4-8 % on every SSL configuration at the
// latency-bound batches) for code that runs next to walls only.
template <int KIND>
__device__ __forceinline__ void blocked_axes(const Params& P, const float x, const float y, uint32_t& bx, uint32_t& by) {
    using K = KC<KIND>;
    static_assert(KIND == RSX_KIND_SSL, "model v2 is defined for the SSL class");
    constexpr float r = K::r_robot;
    const float ax = fabsf(x), ay = fabsf(y);
    const float dxp = ax - P.half_len, dyp = ay - P.ghw;
    const float d2 = fma_(dxp, dxp, dyp * dyp);
    const uint32_t post = __float_as_uint(d2 - r * r) & ~(__float_as_uint(d2) - 1u);   // d2 < r^2 and d2 > 0 (bits(+0) - 1 wraps: sign set)
    const float back = P.half_len + P.gd;
    const uint32_t beyond = __float_as_uint(P.half_len - ax);     // ax > L/2
    const uint32_t in_mouth = __float_as_uint(dyp);               // ay < goal half width
    const uint32_t lt_back = __float_as_uint(ax - back);          // ax < back wall
    const uint32_t inside = in_mouth & lt_back;
    bx = __float_as_uint(((P.half_len + K::margin) - r) - ax) | post |
         (beyond & ((inside & __float_as_uint((back - r) - ax)) | (in_mouth & ~lt_back & __float_as_uint(ax - (back + r)))));
    by = __float_as_uint(((P.half_wid + K::margin) - r) - ay) | post |
         (beyond & ((inside & __float_as_uint((P.ghw - r) - ay)) | (~in_mouth & __float_as_uint(ay - (P.ghw + r)) & lt_back)));
}
template <int KIND>
__device__ __forceinline__ bool wall_shares(const Params& P, const float xa, const float ya, const float xp, const float yp,
                                            const float nx, const float ny, float& wx, float& wy) {
    constexpr float eps = 0.001f;
    uint32_t abx, aby, pbx, pby;
    blocked_axes<KIND>(P, xa - eps * nx, ya - eps * ny, abx, aby);
    blocked_axes<KIND>(P, xp + eps * nx, yp + eps * ny, pbx, pby);
    // blocked body 0, its free partner 1, otherwise 1/2: 1/2 + (partner blocked - body blocked) / 2 (exact)
    wx = fma_((float)(pbx >> 31) - (float)(abx >> 31), 0.5f, 0.5f);
    wy = fma_((float)(pby >> 31) - (float)(aby >> 31), 0.5f, 0.5f);
    return (int32_t)((abx | aby) | (pbx | pby)) < 0;
}

// Model v2, VSS: the same idea without probes or directions (the 1.5 m x 1.3 m field puts a wall pair into the slowest wave of nearly every
// launch: what this costs per touching partner sets the headline step).  A robot is HELD on an axis when its centre lies within 1 mm of
// where the wall clamp (walls<VSS>, a robot's radius) limits that coordinate:
//   y: |y| >= limit - 1 mm — the touch line's limit in the field, the goal's side wall's inside a goal box
//   x: |x| >= limit - 1 mm — the goal line's wall (beside the goal mouth only) in the field, the goal's back wall inside a goal box
// (goal posts hold nothing: a robot slides around them).  h = 1 when held, 0 otherwise.  Shares per axis of a robot - robot pair (a: this
// body, p: its partner): 1/2 + (h_p - h_a) / 2 -> held body 0, its free partner 1, otherwise 1/2 each.  (Which way the contact pushes
// is not asked: a partner that could push a held robot AWAY from its wall would have to stand between the robot and the wall, i.e. be
// held itself.)  Evaluated on the snapshot the sweep reads.  `robot`: the ball and idle lanes publish (0, 0).
template <int KIND>
__device__ __forceinline__ float2 held_axes(const Params& P, const float x, const float y, const bool robot = true) {
    using K = KC<KIND>;
    const float ax = fabsf(x), ay = fabsf(y);
    const bool in_goal = ax > P.half_len;
    // (the same numbers as `(in_goal ? a : b) - r_held`; written as a choice between two differences they are loop invariants of the caller)
    const float yl = in_goal ? P.ghw - K::r_held : P.half_wid - K::r_held;
    const float xl = in_goal ? (P.half_len + P.gd) - K::r_held : P.half_len - K::r_held;
    const bool hy = robot & (ay >= yl);
    const bool hx = robot & (ax >= xl) & (in_goal | (ay >= P.ghw));
    return make_float2(hx ? 1.0f : 0.0f, hy ? 1.0f : 0.0f);
}
// the share differences of a pair from a's point of view, (h_p - h_a) per axis; from p's point of view they are the exact negatives
__device__ __forceinline__ float2 held_diff(const float2 ha, const float2 hp) { return make_float2(hp.x - ha.x, hp.y - ha.y); }

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
// The same with the body's share of the normal impulse and of the de-penetration given PER AXIS (model v2: a robot - robot pair at
// a wall, wall_shares above); the Coulomb limit keeps the nominal share w.  With wx == wy == w every operation is the one respond()
// performs ((ope vn) w, (beta pen) w: same association), so the two agree bit for bit — which lets the kernels keep the short form
// on the hot path and take this one only next to a wall.
__device__ __forceinline__ void respond_axes(const float nx, const float ny, const float pen, const float dvx,
                                             const float dvy, const float wsum, const float ope, const float w,
                                             const float wx, const float wy,
                                             const float kt, const float mu, const float spin_c, const float beta,
                                             float& avx, float& avy, float& apx, float& apy, float& aw) {
    float vn = fma_(dvx, nx, dvy * ny);
    if (vn < 0.0f) {
        float q = ope * vn;
        avx = fma_(q * wx, nx, avx); avy = fma_(q * wy, ny, avy);
        float vt = fma_(dvy, nx, -(dvx * ny)) - wsum;
        float lim = (q * w) * mu;
        float ft = clampf(vt * kt, lim, -lim);
        avx = fma_(-ft, ny, avx); avy = fma_(ft, nx, avy);
        aw = fma_(ft, spin_c, aw);
    }
    float pc = beta * pen;
    apx = fma_(-(pc * wx), nx, apx); apy = fma_(-(pc * wy), ny, apy);
}

// circle - circle pair known to overlap (d2 = squared centre distance)
// rr: a robot - robot pair (per lane): its shares are wall-aware, `wallp` records that an axis was blocked.
// v2w: WAVE-UNIFORM, computed once per sweep by the caller — does any robot of the wave stand at_wall?  If not, no pair can have a
// blocked axis and the partner loop runs v1's instructions, untouched; the wall-aware code hangs off a scalar branch (a divergent
// test here would be structurised in place: its body, exec-mask copies and all, in the middle of the hot loop).
template <int KIND>
__device__ __forceinline__ void contact_response(const Params& P, const Body& o, const float4 oj, const float d2,
                                                 const float rs, const float ope, const float w,
                                                 const float kt, const float mu, const float spin_c,
                                                 const float wsum, const float beta, const float pen2, const bool rr, const bool v2w,
                                                 float& avx, float& avy, float& apx, float& apy, float& aw,
                                                 bool& deep, bool& wallp, const float2 fo = float2{0.0f, 0.0f}, const float2 fj = float2{0.0f, 0.0f}) {
    float dx = oj.x - o.x, dy = oj.y - o.y;
    float d = sqrtf(d2), inv = 1.0f / d;
    const float nx = dx * inv, ny = dy * inv;
    if constexpr (KC<KIND>::held) {
        // VSS: per-axis shares from the held axes of the two bodies (fo: this body's, fj: the partner's, both of this sweep's snapshot);
        // a pair with the ball keeps w on both axes (fma(d, 0, w) == w)
        const float2 hd = held_diff(fo, fj);
        const float hr = rr ? 0.5f : 0.0f;
        respond_axes(nx, ny, rs - d, oj.z - o.vx, oj.w - o.vy, wsum, ope, w, fma_(hd.x, hr, w), fma_(hd.y, hr, w), kt, mu, spin_c, beta, avx, avy, apx, apy, aw);
        deep |= rs - d > pen2;
        return;
    }
    // the hot path is v1's: only a pair with a body within 2 mm of where the wall clamp acts (at_wall: 3 instructions per partner)
    // evaluates the probes — rare, laid out away from the loop
    if (KC<KIND>::wall_aware && __builtin_expect(v2w, 0)) {
        float wx = w, wy = w;
        if constexpr (KC<KIND>::wall_aware)
            if (rr && (at_wall<KIND>(P, o.x, o.y) | at_wall<KIND>(P, oj.x, oj.y))) wallp |= wall_shares<KIND>(P, o.x, o.y, oj.x, oj.y, nx, ny, wx, wy);
        respond_axes(nx, ny, rs - d, oj.z - o.vx, oj.w - o.vy, wsum, ope, w, wx, wy, kt, mu, spin_c, beta, avx, avy, apx, apy, aw);
    } else {
        respond(nx, ny, rs - d, oj.z - o.vx, oj.w - o.vy, wsum, ope, w, kt, mu, spin_c, beta, avx, avy, apx, apy, aw);
    }
    deep |= rs - d > pen2;   // an impact at speed or a jammed pile: the env gets a second sweep
}


// Both sides of ONE touching pair (i, j) in one go — for the layouts in which a lane holds both bodies.  Each side's
// result is bit for bit what contact_response computes from that body's point of view: the geometry of the two
// views differs only by sign (x_i - x_j = -(x_j - x_i) exactly, so d2, d, 1/d, the penetration, the normal speed vn
// and the tangential term fma(dvy, nx, -(dvx ny)) are the same numbers; the normals are each other's negation),
// and a negated operand is free.  Shared: one square root, one division, one normal; per side: the shares w / kt,
// the Coulomb limit, the surface-speed term wsum (whose rounding depends on the side) and the sums.
// ai = {vx, vy, px, py} sums of body i (a robot: no spin), aj the same of body j, awj the spin sum of j (ball only).
// rr: a robot - robot pair: per-axis shares from wall_shares (i's view; j's are the mirror image: the two probes are the same
// points from either side), `wallp` records a blocked axis.
template <int KIND>
__device__ __forceinline__ void contact_pair(const Params& P, const Body& bi, const Body& bj, const float wsum_i, const float wsum_j,
                                             const float rs, const float ope, const float w_i, const float w_j,
                                             const float kt_i, const float kt_j, const float mu, const float spin_c_j,
                                             const float beta, const float pen2, const bool rr, const bool v2w, float* ai, float* aj, float& awj,
                                             bool& deep, bool& wallp) {
    const float dx = bj.x - bi.x, dy = bj.y - bi.y;
    const float d = sqrtf(fma_(dx, dx, dy * dy)), inv = 1.0f / d;
    const float nx = dx * inv, ny = dy * inv, pen = rs - d;           // n: i -> j
    const float dvx = bj.vx - bi.vx, dvy = bj.vy - bi.vy;
    const float vn = fma_(dvx, nx, dvy * ny);
    if (KC<KIND>::held || (KC<KIND>::wall_aware && __builtin_expect(v2w, 0))) {   // (SSL: wave-uniform, see contact_response)
        // model v2: per-axis shares (respond_axes from either side; a pair's shares add up to 1 per axis)
        float wxi = w_i, wyi = w_i, wxj = w_j, wyj = w_j;
        if constexpr (KC<KIND>::wall_aware) {   // SSL: some robot of the wave at a wall
            if (rr && (at_wall<KIND>(P, bi.x, bi.y) | at_wall<KIND>(P, bj.x, bj.y))) {
                wallp |= wall_shares<KIND>(P, bi.x, bi.y, bj.x, bj.y, nx, ny, wxi, wyi);
                wxj = 1.0f - wxi; wyj = 1.0f - wyi;   // {0, 1/2, 1} -> {1, 1/2, 0}: exact
            }
        }
        if constexpr (KC<KIND>::held) {         // VSS: held axes of the two bodies, from the positions this sweep reads (a pair with the ball: w)
            const float2 hd = held_diff(held_axes<KIND>(P, bi.x, bi.y), held_axes<KIND>(P, bj.x, bj.y));
            const float hr = rr ? 0.5f : 0.0f;
            wxi = fma_(hd.x, hr, w_i); wyi = fma_(hd.y, hr, w_i);
            wxj = fma_(-hd.x, hr, w_j); wyj = fma_(-hd.y, hr, w_j);
        }
        if (vn < 0.0f) {
            const float vt0 = fma_(dvy, nx, -(dvx * ny));
            const float q0 = ope * vn;
            {   // body i
                ai[0] = fma_(q0 * wxi, nx, ai[0]); ai[1] = fma_(q0 * wyi, ny, ai[1]);
                const float lim = (q0 * w_i) * mu;
                const float ft = clampf((vt0 - wsum_i) * kt_i, lim, -lim);
                ai[0] = fma_(-ft, ny, ai[0]); ai[1] = fma_(ft, nx, ai[1]);
            }
            {   // body j: normal -n
                aj[0] = fma_(q0 * wxj, -nx, aj[0]); aj[1] = fma_(q0 * wyj, -ny, aj[1]);
                const float lim = (q0 * w_j) * mu;
                const float ft = clampf((vt0 - wsum_j) * kt_j, lim, -lim);
                aj[0] = fma_(ft, ny, aj[0]); aj[1] = fma_(ft, -nx, aj[1]);
                awj = fma_(ft, spin_c_j, awj);
            }
        }
        const float pc = beta * pen;
        ai[2] = fma_(-(pc * wxi), nx, ai[2]); ai[3] = fma_(-(pc * wyi), ny, ai[3]);
        aj[2] = fma_(pc * wxj, nx, aj[2]); aj[3] = fma_(pc * wyj, ny, aj[3]);
    } else {
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
    }
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
