// rsx_params.hpp — model constants of the step engine.
//
// Two tiers, because scalar registers are the scarce resource of these kernels (a wave has
// ~100 SGPRs and the step needs ~15 pointers/ints besides the constants):
//   * KC<KIND> / TC<TASK>: everything that depends only on the robot class or the task is a
//     compile-time literal (folded into the VALU instructions as 32-bit immediates);
//   * Params: what depends on run-time choices (field type, time step, batch, seeds) — one POD
//     block passed as the kernel argument.
// Both are derived from the same double-precision expressions (ModelD below), so the literal
// a kernel uses and the value the host reports can never drift apart.
//
// Provenance tags: [ref] = literal visible in the reference tree (file:line given),
// [build] = chosen by this project because the value only exists inside rc-robosim, which is
// not part of /root/reference.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "rsx.h"

namespace rsx {

constexpr int MAX_ROBOTS = 22;
// internal state rows behind the get_state() rows: ball vertical velocity, ball spin (rad/s)
constexpr int X_ROWS = 2;
constexpr double PI_D = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------------
// per-robot-class physical constants (double)
// ---------------------------------------------------------------------------------------------
template <int KIND> struct ModelD;

template <> struct ModelD<RSX_KIND_VSS> {
    static constexpr double r_ball = 0.0215;      // [ref] Render/ball.py:6
    static constexpr double r_robot = 0.0375;     // [ref] vss_gym_base.py:57
    static constexpr double lever = 0.04;         // [ref] vss_gym_base.py:58 (radius + wheel thickness)
    static constexpr double r_wheel = 0.026;      // [build]
    static constexpr double rpm = 440.0;          // [build]
    static constexpr double dck = 0.0, kick_t = 0.0, kick_w = 0.0;
    static constexpr double wheel_deg[4] = {90.0, 270.0, 0.0, 0.0};  // [build] left / right
    static constexpr double m_robot = 0.18, m_ball = 0.046;          // [build]
    static constexpr double a_lin = 8.0, a_lat = 20.0, a_ang = 300.0, mu_g = 0.3;  // [build]
    static constexpr double e_rr = 0.1, e_rb = 0.3, e_wb = 0.6, e_wr = 0.1;        // [build]
    static constexpr double mu_rr = 0.2, mu_rb = 0.35, mu_wb = 0.3, spin_dec = 30.0;  // [build] Coulomb friction in contacts, spin deceleration (rad/s^2)
    static constexpr double margin = 0.0;
    static constexpr int rs = 6, cmd_dim = 2;     // Entities/Frame.py:27, rsim.py:93
};

template <> struct ModelD<RSX_KIND_SSL> {
    static constexpr double r_ball = 0.0215;
    static constexpr double r_robot = 0.09;       // [ref] ssl_gym_base.py:58
    static constexpr double lever = 0.09;
    static constexpr double r_wheel = 0.02475;    // [build]
    static constexpr double rpm = 160.0 * 60.0 / (2.0 * PI_D);  // 160 rad/s [ref] static_defenders.py:71
    static constexpr double dck = 0.073, kick_t = 0.005, kick_w = 0.08;   // [build] kicker
    static constexpr double wheel_deg[4] = {60.0, 135.0, 225.0, 300.0};   // [build] omni wheels
    static constexpr double m_robot = 2.2, m_ball = 0.046;
    static constexpr double a_lin = 5.0, a_lat = 0.0, a_ang = 50.0, mu_g = 0.4;
    static constexpr double e_rr = 0.1, e_rb = 0.2, e_wb = 0.5, e_wr = 0.1;
    static constexpr double mu_rr = 0.2, mu_rb = 0.35, mu_wb = 0.3, spin_dec = 30.0;
    static constexpr double margin = 0.3;
    static constexpr int rs = 11, cmd_dim = 8;    // Entities/Frame.py:62, rsim.py:130
};

constexpr double BETA_D = 0.8, GRAV_D = 9.81;     // [build] de-penetration share, gravity

// float literals the kernels use
template <int KIND>
struct KC {
    using D = ModelD<KIND>;
    static constexpr float r_robot = (float)D::r_robot, r_ball = (float)D::r_ball;
    static constexpr float margin = (float)D::margin;
    static constexpr float rs_rr = (float)(2.0 * D::r_robot);
    static constexpr float rs_rr2 = (float)((2.0 * D::r_robot) * (2.0 * D::r_robot));
    static constexpr float rs_rb = (float)(D::r_robot + D::r_ball);
    static constexpr float rs_rb2 = (float)((D::r_robot + D::r_ball) * (D::r_robot + D::r_ball));
    static constexpr double imr = 1.0 / D::m_robot, imb = 1.0 / D::m_ball;
    static constexpr float w_rr = 0.5f;
    // model v2 [build], DESIGN.md 4: robot - robot pairs at a wall get per-axis shares.  SSL (rsx_body.hpp: wall_shares): blocked axes
    // found by probing, piles at a wall get a third and a fourth sweep.  VSS (rsx_body.hpp: held_axes): a robot within 1 mm of where the
    // wall clamp limits a coordinate is held on that axis; two sweeps.
    static constexpr bool wall_aware = KIND == RSX_KIND_SSL;
    static constexpr bool held = KIND == RSX_KIND_VSS;
    static constexpr float r_held = (float)(D::r_robot + 0.001);
    static constexpr float w_rb_r = (float)(imr / (imr + imb)), w_rb_b = (float)(imb / (imr + imb));
    static constexpr float ope_rr = (float)(1.0 + D::e_rr), ope_rb = (float)(1.0 + D::e_rb);
    static constexpr float e_wb = (float)D::e_wb, e_wr = (float)D::e_wr, beta = (float)BETA_D;
    static constexpr float w_max = (float)(D::rpm / 60.0 * 2.0 * PI_D);
    static constexpr float r_wheel = (float)D::r_wheel;
    static constexpr float half_rw = (float)(D::r_wheel * 0.5);
    static constexpr float rw_2b = (float)(D::r_wheel / (2.0 * D::lever));
    static constexpr float inv_rw = (float)(1.0 / D::r_wheel);
    static constexpr float e_ground = 0.5f, vz_min = 0.2f, robot_h = 0.15f;   // [build]
    static constexpr float dck_rb = (float)(D::dck + D::r_ball);
    static constexpr float half_kw = (float)(D::kick_w / 2), ir_tol = 0.025f;
    static constexpr float drib_vmax = 1.0f, drib_vmax2 = 1.0f;
    static constexpr float deg2rad = (float)(PI_D / 180.0), rad2deg = (float)(180.0 / PI_D);
    // Coulomb friction in contacts.  Tangential effective mass: robots are yaw-controlled by their
    // motors (no torque from contacts), the ball is a solid sphere (I = 2/5 m r^2: the contact point
    // adds r^2 / I = 2.5 / m), so 1/m_t = 1/m_r + 3.5/m_b (robot-ball), 2/m_r (robot-robot),
    // 3.5/m_b (wall-ball).  kt_* = m_t / m_body, spin_c = spin per unit of tangential velocity change.
    static constexpr double mt_rb = 1.0 / (imr + 3.5 * imb);
    static constexpr float mu_rr = (float)D::mu_rr, mu_rb = (float)D::mu_rb, mu_wb = (float)D::mu_wb;
    static constexpr float kt_rr = 0.5f, kt_rb_r = (float)(mt_rb * imr), kt_rb_b = (float)(mt_rb * imb);
    static constexpr float kw = (float)(2.0 / 7.0), spin_c = (float)(2.5 / D::r_ball);
    static constexpr float ope_wb = (float)(1.0 + D::e_wb), dck = (float)D::dck;
    // overlap beyond which an env gets the second contact sweep of a sub-step [build]
    static constexpr float pen2 = 0.005f;
};

// per-task literals
template <int TASK> struct TC;
template <> struct TC<RSX_TASK_VSS_V0> {  // normalisers: vss_gym_base.py:52-58
    using D = ModelD<RSX_KIND_VSS>;
    static constexpr double max_v_d = (D::rpm / 60.0) * 2.0 * PI_D * D::r_wheel;
    static constexpr double max_w_d = (max_v_d / 0.04) * (180.0 / PI_D);
    static constexpr float max_v = (float)max_v_d, inv_max_v = (float)(1.0 / max_v_d);
    static constexpr float inv_max_w = (float)(1.0 / max_w_d);
    static constexpr float deadzone = 0.05f;                         // vss_gym.py:73
    static constexpr float inv_en_scale = 0.0f;                      // (SSL tasks only)
    static constexpr int info_dim = 6, act_dim = 2, max_steps = 1200; // rsoccer_gym/__init__.py:4
};
// the four SSL hardware-challenge tasks share the speed caps 2.5 m/s / 10 rad/s
// (static_defenders.py:76-77, dribbling.py:66-67, contested_possession.py:65-66, pass_endurance.py:72-73)
struct TCSslBase {
    static constexpr double max_v_d = 2.5, max_w_d = 10.0;
    static constexpr float max_v = 2.5f, inv_max_v = (float)(1.0 / 2.5), inv_max_w = (float)(1.0 / 10.0);
    static constexpr float deadzone = 0.0f;
};
template <> struct TC<RSX_TASK_SSL_STATIC_DEFENDERS> : TCSslBase {
    static constexpr float inv_en_scale = (float)(1.0 / (160.0 * 4.0 * 1000.0));  // static_defenders.py:71-73
    static constexpr int info_dim = 8, act_dim = 5, max_steps = 1000;  // rsoccer_gym/__init__.py:11
    static constexpr int n_blue = 1, n_yellow = -1;                    // -1: any
};
template <> struct TC<RSX_TASK_SSL_DRIBBLING> : TCSslBase {
    static constexpr float inv_en_scale = 0.0f;
    static constexpr int info_dim = 1, act_dim = 4, max_steps = 4800;  // rsoccer_gym/__init__.py:17
    static constexpr int n_blue = 1, n_yellow = 4;                     // dribbling.py:47-48
};
template <> struct TC<RSX_TASK_SSL_CONTESTED> : TCSslBase {
    static constexpr float inv_en_scale = (float)(1.0 / (160.0 * 4.0 * 1200.0));  // contested_possession.py:60-62
    static constexpr int info_dim = 9, act_dim = 5, max_steps = 1200;  // rsoccer_gym/__init__.py:23
    static constexpr int n_blue = 1, n_yellow = 1;                     // contested_possession.py:43-44
};
template <> struct TC<RSX_TASK_SSL_PASS_ENDURANCE> : TCSslBase {
    static constexpr float inv_en_scale = 0.0f;
    static constexpr int info_dim = 2, act_dim = 3, max_steps = 1200;  // rsoccer_gym/__init__.py:29
    static constexpr int n_blue = 2, n_yellow = 0;                     // pass_endurance.py:48-49
};

template <> struct TC<RSX_TASK_SSL_SCRIMMAGE> : TCSslBase {   // both line-ups run this variant
    static constexpr float inv_en_scale = 0.0f;
    static constexpr int info_dim = 2, act_dim = 4 /* per robot */, max_steps = 1200;
    static constexpr int n_blue = -1, n_yellow = -1;
};

// ---------------------------------------------------------------------------------------------
// run-time block (kernel argument)
// ---------------------------------------------------------------------------------------------
struct Params {
    int kind, n_blue, n_yellow, n_robots, n_sub, state_dim, num_envs;
    int row_stride;   // floats between consecutive rows of the [rows][B] arrays state / cmds / aux: num_envs + a pad (rsx_api.hip: row_pad_for)
    // sub-step and field dependent
    float h, h_deg, a_lin_h, a_lin_h2, a_lat_h, a_ang_h, mu_g_dt, g_h, drib_gain, spin_dec_dt;
    float half_len, half_wid, ghw, gd;
    // omni-wheel kinematics (SSL; used once per step)
    float ws[4], wc[4], pinv[3][4];
    // task
    int task, obs_dim, max_steps;
    uint32_t key0, key1, env_id_base;
    uint32_t tick_base;   // step launches of this handle since attach: key of the per-step draws
    float inv_max_pos, hl_goal, inv_len_cm, inv_dt;
    float pen_x, half_pen_wid, inv_bd_scale, inv_bg_scale;
    float pl_xlo, pl_xspan, pl_ylo, pl_yspan, pl_min_d2;
    float ou_theta_dt, ou_sig_sqdt;
    float sc_sx, sc_sy, sc_j, sc_jb;   // scrimmage line-up: grid spacing, robot jitter, ball jitter
};

struct HostModel {
    double field[RSX_FIELD_PARAMS];  // Entities/Field.py:5-21 order
    double dt;                       // seconds per step()
    int rs, cmd_dim, act_dim, info_dim;
};

template <int KIND>
inline int derive_model_k(int field_type, int ts_ms, Params& P, HostModel& M) {
    using D = ModelD<KIND>;
    double* f = M.field;
    if (KIND == RSX_KIND_VSS) {
        switch (field_type) {
            case 0: f[0] = 1.5; f[1] = 1.3; f[2] = 0.15; f[3] = 0.7; f[4] = 0.4; f[5] = 0.1; break;  // [ref] Render/field.py:190-199
            case 1: f[0] = 2.2; f[1] = 1.8; f[2] = 0.15; f[3] = 0.8; f[4] = 0.4; f[5] = 0.15; break; // [build] 5v5
            default: return -1;
        }
    } else {
        switch (field_type) {
            case 0: f[0] = 9.0; f[1] = 6.0; f[2] = 1.0; f[3] = 2.0; f[4] = 1.0; f[5] = 0.18; break;   // [ref] Render/field.py:253-262
            case 1: f[0] = 12.0; f[1] = 9.0; f[2] = 1.8; f[3] = 3.6; f[4] = 1.8; f[5] = 0.18; break;  // [build] division A
            case 2: f[0] = 6.0; f[1] = 4.0; f[2] = 0.8; f[3] = 2.0; f[4] = 1.0; f[5] = 0.18; break;   // [build] hw challenge
            default: return -1;
        }
    }
    f[6] = D::r_ball; f[7] = D::dck; f[8] = D::kick_t; f[9] = D::kick_w;
    for (int k = 0; k < 4; ++k) f[10 + k] = D::wheel_deg[k];
    f[14] = D::r_robot; f[15] = D::r_wheel; f[16] = D::rpm;
    M.rs = D::rs; M.cmd_dim = D::cmd_dim;
    P.n_sub = (ts_ms + 4) / 5;                  // 5 ms sub-steps [build]
    M.dt = ts_ms * 0.001;
    const double h = P.n_sub ? M.dt / P.n_sub : 0.0;
    P.h = (float)h;
    P.h_deg = (float)(h * (180.0 / PI_D));      // heading is integrated in degrees, the wire unit
    P.a_lin_h = (float)(D::a_lin * h); P.a_lin_h2 = (float)((D::a_lin * h) * (D::a_lin * h));
    P.a_lat_h = (float)(D::a_lat * h); P.a_ang_h = (float)(D::a_ang * h);
    P.mu_g_dt = (float)(D::mu_g * (ts_ms * 0.001)); P.g_h = (float)(GRAV_D * h);
    P.spin_dec_dt = (float)(D::spin_dec * (ts_ms * 0.001));
    P.drib_gain = (float)(h > 0 ? 0.5 / h : 0.0);
    P.half_len = (float)(f[0] / 2); P.half_wid = (float)(f[1] / 2);
    P.ghw = (float)(f[4] / 2); P.gd = (float)f[5];
    double ang[4];
    for (int k = 0; k < 4; ++k) {
        ang[k] = f[10 + k] * PI_D / 180.0;
        P.ws[k] = (float)std::sin(ang[k]); P.wc[k] = (float)std::cos(ang[k]);
    }
    if (KIND == RSX_KIND_SSL) {
        // Wheel surface speed_k = -sin(a_k) vx + cos(a_k) vy + R w  (robot frame).
        // The wheel-speed command mode needs the least-squares inverse (J^T J)^-1 J^T.
        double J[4][3], A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Ai[3][3];
        for (int k = 0; k < 4; ++k) { J[k][0] = -std::sin(ang[k]); J[k][1] = std::cos(ang[k]); J[k][2] = D::r_robot; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 4; ++k) A[i][j] += J[k][i] * J[k][j];
        const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1])
                         - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0])
                         + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        Ai[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det;
        Ai[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
        Ai[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
        Ai[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
        Ai[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
        Ai[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
        Ai[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
        Ai[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
        Ai[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
        for (int i = 0; i < 3; ++i) for (int k = 0; k < 4; ++k) {
            double s = 0;
            for (int j = 0; j < 3; ++j) s += Ai[i][j] * J[k][j];
            P.pinv[i][k] = (float)s;
        }
    }
    return 0;
}

// Returns 0 on success.  Fills the physics part of P and the field table.
inline int derive_model(int kind, int field_type, int nb, int ny, int ts_ms, int num_envs,
                        Params& P, HostModel& M) {
    std::memset(&P, 0, sizeof(P));
    std::memset(&M, 0, sizeof(M));
    if (kind != RSX_KIND_VSS && kind != RSX_KIND_SSL) return -1;
    if (nb < 0 || ny < 0 || nb + ny < 1 || nb + ny > MAX_ROBOTS || ts_ms < 0 || num_envs < 1) return -1;
    P.kind = kind; P.n_blue = nb; P.n_yellow = ny; P.n_robots = nb + ny; P.num_envs = num_envs; P.row_stride = num_envs;
    const int rc = kind == RSX_KIND_VSS ? derive_model_k<RSX_KIND_VSS>(field_type, ts_ms, P, M)
                                        : derive_model_k<RSX_KIND_SSL>(field_type, ts_ms, P, M);
    if (rc) return rc;
    P.state_dim = 5 + M.rs * P.n_robots;
    return 0;
}

// Task constants.  Returns 0 on success, -1 when the task does not fit the simulator.
inline int derive_task(int task, uint64_t seed, uint64_t env_id_base, int max_steps,
                       HostModel& M, Params& P) {
    const double* f = M.field;
    if (task == RSX_TASK_VSS_V0) {
        using T = TC<RSX_TASK_VSS_V0>;
        if (P.kind != RSX_KIND_VSS || P.n_blue < 1) return -1;
        P.obs_dim = 4 + 7 * P.n_blue + 5 * P.n_yellow;   // vss_gym.py:64-67
        M.act_dim = T::act_dim; M.info_dim = T::info_dim;
        P.max_steps = max_steps > 0 ? max_steps : T::max_steps;
    } else if (task == RSX_TASK_SSL_STATIC_DEFENDERS) {
        using T = TC<RSX_TASK_SSL_STATIC_DEFENDERS>;
        if (P.kind != RSX_KIND_SSL || P.n_blue != 1) return -1;
        P.obs_dim = 4 + 8 * P.n_blue + 2 * P.n_yellow;   // static_defenders.py:54-56
        M.act_dim = T::act_dim; M.info_dim = T::info_dim;
        P.max_steps = max_steps > 0 ? max_steps : T::max_steps;
    } else if (task == RSX_TASK_SSL_DRIBBLING) {
        using T = TC<RSX_TASK_SSL_DRIBBLING>;
        if (P.kind != RSX_KIND_SSL || P.n_blue != T::n_blue || P.n_yellow != T::n_yellow) return -1;
        P.obs_dim = 5 + 8 * P.n_blue + 2 * P.n_yellow;   // dribbling.py:52
        M.act_dim = T::act_dim; M.info_dim = T::info_dim;
        P.max_steps = max_steps > 0 ? max_steps : T::max_steps;
    } else if (task == RSX_TASK_SSL_CONTESTED) {
        using T = TC<RSX_TASK_SSL_CONTESTED>;
        if (P.kind != RSX_KIND_SSL || P.n_blue != T::n_blue || P.n_yellow != T::n_yellow) return -1;
        P.obs_dim = 4 + 8 * P.n_blue + 2 * P.n_yellow;   // contested_possession.py:48
        M.act_dim = T::act_dim; M.info_dim = T::info_dim;
        P.max_steps = max_steps > 0 ? max_steps : T::max_steps;
    } else if (task == RSX_TASK_SSL_PASS_ENDURANCE) {
        using T = TC<RSX_TASK_SSL_PASS_ENDURANCE>;
        if (P.kind != RSX_KIND_SSL || P.n_blue != T::n_blue || P.n_yellow != T::n_yellow) return -1;
        P.obs_dim = 4 + 6 * P.n_blue;                    // pass_endurance.py:55
        M.act_dim = T::act_dim; M.info_dim = T::info_dim;
        P.max_steps = max_steps > 0 ? max_steps : T::max_steps;
    } else if (task == RSX_TASK_SSL_SCRIMMAGE || task == RSX_TASK_SSL_SCRIMMAGE_CROWDED) {
        using T = TC<RSX_TASK_SSL_SCRIMMAGE>;
        if (P.kind != RSX_KIND_SSL || P.n_robots < 1) return -1;
        P.obs_dim = 2 + 2 * P.n_robots;
        M.act_dim = T::act_dim * P.n_robots; M.info_dim = T::info_dim;
        P.max_steps = max_steps > 0 ? max_steps : T::max_steps;
        if (task == RSX_TASK_SSL_SCRIMMAGE) {   // jittered grid over the field: >= 0.2 m apart by construction
            const double sx = f[0] / 8.0, sy = f[1] / 5.0, j = 0.3 * (sx < sy ? sx : sy);
            P.sc_sx = (float)sx; P.sc_sy = (float)sy; P.sc_j = (float)j; P.sc_jb = 0.1f;
        } else {                                // the same grid packed around the ball
            P.sc_sx = 0.25f; P.sc_sy = 0.25f; P.sc_j = 0.02f; P.sc_jb = 0.02f;
        }
    } else {
        return -1;
    }
    P.task = task;
    P.key0 = (uint32_t)seed; P.key1 = (uint32_t)(seed >> 32); P.env_id_base = (uint32_t)env_id_base;
    const double max_pos = std::fmax(f[1] / 2, f[0] / 2 + f[2]);   // vss_gym_base.py:52-54
    P.inv_max_pos = (float)(1.0 / max_pos);
    P.hl_goal = (float)(f[0] / 2.0 + f[5]); P.inv_len_cm = (float)(1.0 / (f[0] * 100.0)); // vss_gym.py:261-262
    P.inv_dt = (float)(M.dt > 0 ? 1.0 / M.dt : 0.0);
    P.pen_x = (float)(f[0] / 2 - f[2]); P.half_pen_wid = (float)(f[3] / 2);
    P.inv_bd_scale = (float)(1.0 / std::sqrt(f[1] * f[1] + (f[0] / 2) * (f[0] / 2)));   // static_defenders.py:65
    P.inv_bg_scale = (float)(1.0 / (std::sqrt((f[1] / 2) * (f[1] / 2) + (f[0] / 2) * (f[0] / 2)) / 4.0)); // :66-68
    if (task == RSX_TASK_VSS_V0) {          // vss_gym.py:199-206,211
        P.pl_xlo = (float)(-(f[0] / 2) + 0.1); P.pl_xspan = (float)((f[0] / 2 - 0.1) - (-(f[0] / 2) + 0.1));
        P.pl_min_d2 = (float)(0.1 * 0.1);
    } else if (task == RSX_TASK_SSL_CONTESTED) {   // contested_possession.py:210-211: the opponent's spot
        P.pl_xlo = (float)f[2]; P.pl_xspan = (float)((f[0] / 2 - f[2]) - f[2]);
        P.pl_min_d2 = 0.0f;
    } else {                                // static_defenders.py:221-225,239
        P.pl_xlo = 0.2f; P.pl_xspan = (float)((f[0] / 2 - 0.1) - 0.2);
        P.pl_min_d2 = (float)(0.2 * 0.2);
    }
    P.pl_ylo = (float)(-(f[1] / 2) + 0.1); P.pl_yspan = (float)((f[1] / 2 - 0.1) - (-(f[1] / 2) + 0.1));
    if (task == RSX_TASK_SSL_CONTESTED) { P.pl_ylo = (float)(-(f[3] / 2)); P.pl_yspan = (float)f[3]; }
    P.ou_theta_dt = (float)(0.17 * M.dt);            // Utils/Utils.py:6,17
    P.ou_sig_sqdt = (float)(0.5 * std::sqrt(M.dt));  // Utils/Utils.py:8,18
    return 0;
}

}  // namespace rsx
