// rsx_params.hpp — model constants of the step engine: field tables, robot/ball dynamics and
// the per-task normalisers, derived on the host in double precision and handed to the kernels
// as one POD block (kernel argument, lives in SGPRs / scalar cache).
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
constexpr double PI_D = 3.14159265358979323846;

struct Params {
    // ---- shape ----
    int kind, n_blue, n_yellow, n_robots, n_sub, rs, state_dim, cmd_dim, num_envs;
    // ---- geometry ----
    float h, half_len, half_wid, ghw, gd, margin, r_robot, r_ball;
    float rs_rr, rs_rr2, rs_rb, rs_rb2;
    // ---- contacts ----
    float w_rr, w_rb_r, w_rb_b, ope_rr, ope_rb, e_wb, e_wr, beta;
    // ---- actuation ----
    float w_max, half_rw, rw_2b, inv_rw, r_wheel;
    float a_lin_h, a_lin_h2, a_lat_h, a_ang_h, mu_g_h, g_h, e_ground, vz_min, robot_h;
    float dck_rb, half_kw, ir_tol, drib_gain, drib_vmax, drib_vmax2;
    float ws[4], wc[4], pinv[3][4];
    float deg2rad, rad2deg, pi, two_pi;
    // ---- task ----
    int task, obs_dim, act_dim, info_dim, max_steps;
    uint32_t key0, key1, env_id_base;
    float max_pos, inv_max_pos, max_v, inv_max_v, inv_max_w, deadzone;
    float hl_goal, inv_len_cm, inv_dt;
    float pen_x, half_pen_wid, inv_bd_scale, inv_bg_scale, inv_en_scale;
    float pl_xlo, pl_xspan, pl_ylo, pl_yspan, pl_min_d2;
    float ou_theta_dt, ou_sig_sqdt;
};

struct HostModel {
    double field[RSX_FIELD_PARAMS];  // Entities/Field.py:5-21 order
    double dt;                       // seconds per step()
};

// Returns 0 on success.  Fills the physics part of P and the field table.
inline int derive_model(int kind, int field_type, int nb, int ny, int ts_ms, int num_envs,
                        Params& P, HostModel& M) {
    std::memset(&P, 0, sizeof(P));
    std::memset(&M, 0, sizeof(M));
    if (kind != RSX_KIND_VSS && kind != RSX_KIND_SSL) return -1;
    if (nb < 0 || ny < 0 || nb + ny < 1 || nb + ny > MAX_ROBOTS || ts_ms < 0 || num_envs < 1) return -1;
    double* f = M.field;
    double m_robot, m_ball = 0.046, a_lin, a_lat, a_ang, mu_g, e_rr = 0.1, e_rb, e_wb, e_wr = 0.1;
    double margin, lever;
    if (kind == RSX_KIND_VSS) {
        switch (field_type) {
            case 0: f[0] = 1.5; f[1] = 1.3; f[2] = 0.15; f[3] = 0.7; f[4] = 0.4; f[5] = 0.1; break;  // [ref] Render/field.py:190-199
            case 1: f[0] = 2.2; f[1] = 1.8; f[2] = 0.15; f[3] = 0.8; f[4] = 0.4; f[5] = 0.15; break; // [build] 5v5
            default: return -1;
        }
        f[6] = 0.0215;                          // [ref] Render/ball.py:6
        f[10] = 90.0; f[11] = 270.0;            // [build] left / right wheel
        f[14] = 0.0375;                         // [ref] vss_gym_base.py:57
        f[15] = 0.026; f[16] = 440.0;           // [build]
        m_robot = 0.18; a_lin = 8.0; a_lat = 20.0; a_ang = 300.0; mu_g = 0.3;   // [build]
        e_rb = 0.3; e_wb = 0.6; margin = 0.0;
        lever = 0.04;                           // [ref] vss_gym_base.py:58
    } else {
        switch (field_type) {
            case 0: f[0] = 9.0; f[1] = 6.0; f[2] = 1.0; f[3] = 2.0; f[4] = 1.0; f[5] = 0.18; break;   // [ref] Render/field.py:253-262
            case 1: f[0] = 12.0; f[1] = 9.0; f[2] = 1.8; f[3] = 3.6; f[4] = 1.8; f[5] = 0.18; break;  // [build] division A
            case 2: f[0] = 6.0; f[1] = 4.0; f[2] = 0.8; f[3] = 2.0; f[4] = 1.0; f[5] = 0.18; break;   // [build] hw challenge
            default: return -1;
        }
        f[6] = 0.0215;
        f[7] = 0.073; f[8] = 0.005; f[9] = 0.08;                    // [build] kicker
        f[10] = 60.0; f[11] = 135.0; f[12] = 225.0; f[13] = 300.0;  // [build] omni wheels
        f[14] = 0.09;                           // [ref] ssl_gym_base.py:58
        f[15] = 0.02475;                        // [build]
        f[16] = 160.0 * 60.0 / (2.0 * PI_D);    // 160 rad/s [ref] static_defenders.py:71
        m_robot = 2.2; a_lin = 5.0; a_lat = 0.0; a_ang = 50.0; mu_g = 0.4;      // [build]
        e_rb = 0.2; e_wb = 0.5; margin = 0.3;
        lever = 0.09;
    }
    const double beta = 0.8, grav = 9.81;
    const double r_robot = f[14], r_ball = f[6], r_wheel = f[15];
    P.kind = kind; P.n_blue = nb; P.n_yellow = ny; P.n_robots = nb + ny; P.num_envs = num_envs;
    P.rs = kind == RSX_KIND_VSS ? 6 : 11;
    P.state_dim = 5 + P.rs * P.n_robots;
    P.cmd_dim = kind == RSX_KIND_VSS ? 2 : 8;
    P.n_sub = (ts_ms + 4) / 5;                  // 5 ms sub-steps [build]
    M.dt = ts_ms * 0.001;
    const double h = P.n_sub ? M.dt / P.n_sub : 0.0;
    P.h = (float)h;
    P.half_len = (float)(f[0] / 2); P.half_wid = (float)(f[1] / 2);
    P.ghw = (float)(f[4] / 2); P.gd = (float)f[5]; P.margin = (float)margin;
    P.r_robot = (float)r_robot; P.r_ball = (float)r_ball;
    P.rs_rr = (float)(2.0 * r_robot); P.rs_rr2 = (float)((2.0 * r_robot) * (2.0 * r_robot));
    P.rs_rb = (float)(r_robot + r_ball); P.rs_rb2 = (float)((r_robot + r_ball) * (r_robot + r_ball));
    const double imr = 1.0 / m_robot, imb = 1.0 / m_ball;
    P.w_rr = 0.5f; P.w_rb_r = (float)(imr / (imr + imb)); P.w_rb_b = (float)(imb / (imr + imb));
    P.ope_rr = (float)(1.0 + e_rr); P.ope_rb = (float)(1.0 + e_rb);
    P.e_wb = (float)e_wb; P.e_wr = (float)e_wr; P.beta = (float)beta;
    const double w_max = f[16] / 60.0 * 2.0 * PI_D;
    P.w_max = (float)w_max; P.r_wheel = (float)r_wheel;
    P.half_rw = (float)(r_wheel * 0.5); P.rw_2b = (float)(r_wheel / (2.0 * lever));
    P.inv_rw = (float)(1.0 / r_wheel);
    P.a_lin_h = (float)(a_lin * h); P.a_lin_h2 = (float)((a_lin * h) * (a_lin * h));
    P.a_lat_h = (float)(a_lat * h); P.a_ang_h = (float)(a_ang * h);
    P.mu_g_h = (float)(mu_g * h); P.g_h = (float)(grav * h);
    P.e_ground = 0.5f; P.vz_min = 0.2f; P.robot_h = 0.15f;
    P.dck_rb = (float)(f[7] + r_ball); P.half_kw = (float)(f[9] / 2); P.ir_tol = 0.01f;
    P.drib_gain = (float)(h > 0 ? 0.5 / h : 0.0);
    P.drib_vmax = 1.0f; P.drib_vmax2 = 1.0f;
    double ang[4];
    for (int k = 0; k < 4; ++k) {
        ang[k] = f[10 + k] * PI_D / 180.0;
        P.ws[k] = (float)std::sin(ang[k]); P.wc[k] = (float)std::cos(ang[k]);
    }
    if (kind == RSX_KIND_SSL) {
        // Wheel surface speed_k = -sin(a_k) vx + cos(a_k) vy + R w  (robot frame).
        // The wheel-speed command mode needs the least-squares inverse (J^T J)^-1 J^T.
        double J[4][3], A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Ai[3][3];
        for (int k = 0; k < 4; ++k) { J[k][0] = -std::sin(ang[k]); J[k][1] = std::cos(ang[k]); J[k][2] = r_robot; }
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
    P.deg2rad = (float)(PI_D / 180.0); P.rad2deg = (float)(180.0 / PI_D);
    P.pi = (float)PI_D; P.two_pi = (float)(2.0 * PI_D);
    return 0;
}

// Task constants.  Returns 0 on success, -1 when the task does not fit the simulator.
inline int derive_task(int task, uint64_t seed, uint64_t env_id_base, int max_steps,
                       const HostModel& M, Params& P) {
    const double* f = M.field;
    if (task == RSX_TASK_VSS_V0) {
        if (P.kind != RSX_KIND_VSS || P.n_blue < 1) return -1;
        P.obs_dim = 4 + 7 * P.n_blue + 5 * P.n_yellow; P.act_dim = 2; P.info_dim = 6;   // vss_gym.py:64-67
        P.max_steps = max_steps > 0 ? max_steps : 1200;                                 // rsoccer_gym/__init__.py:4
    } else if (task == RSX_TASK_SSL_STATIC_DEFENDERS) {
        if (P.kind != RSX_KIND_SSL || P.n_blue != 1) return -1;
        P.obs_dim = 4 + 8 * P.n_blue + 2 * P.n_yellow; P.act_dim = 5; P.info_dim = 8;   // static_defenders.py:54-56
        P.max_steps = max_steps > 0 ? max_steps : 1000;                                 // rsoccer_gym/__init__.py:11
    } else {
        return -1;
    }
    P.task = task;
    P.key0 = (uint32_t)seed; P.key1 = (uint32_t)(seed >> 32); P.env_id_base = (uint32_t)env_id_base;
    // normalisers: vss_gym_base.py:52-58, ssl_gym_base.py:53-59
    const double max_pos = std::fmax(f[1] / 2, f[0] / 2 + f[2]);
    double max_v = (f[16] / 60.0) * 2.0 * PI_D * f[15];
    double max_w = (max_v / (P.kind == RSX_KIND_VSS ? 0.04 : 0.095)) * (180.0 / PI_D);
    if (task == RSX_TASK_SSL_STATIC_DEFENDERS) { max_v = 2.5; max_w = 10.0; }           // static_defenders.py:76-77
    P.max_pos = (float)max_pos; P.inv_max_pos = (float)(1.0 / max_pos);
    P.max_v = (float)max_v; P.inv_max_v = (float)(1.0 / max_v); P.inv_max_w = (float)(1.0 / max_w);
    P.deadzone = 0.05f;                                                                  // vss_gym.py:73
    P.hl_goal = (float)(f[0] / 2.0 + f[5]); P.inv_len_cm = (float)(1.0 / (f[0] * 100.0)); // vss_gym.py:261-262
    P.inv_dt = (float)(M.dt > 0 ? 1.0 / M.dt : 0.0);
    P.pen_x = (float)(f[0] / 2 - f[2]); P.half_pen_wid = (float)(f[3] / 2);
    P.inv_bd_scale = (float)(1.0 / std::sqrt(f[1] * f[1] + (f[0] / 2) * (f[0] / 2)));   // static_defenders.py:65
    P.inv_bg_scale = (float)(1.0 / (std::sqrt((f[1] / 2) * (f[1] / 2) + (f[0] / 2) * (f[0] / 2)) / 4.0)); // :66-68
    P.inv_en_scale = (float)(1.0 / (160.0 * 4.0 * 1000.0));                              // :71-73
    if (task == RSX_TASK_VSS_V0) {          // vss_gym.py:199-206,211
        P.pl_xlo = (float)(-(f[0] / 2) + 0.1); P.pl_xspan = (float)((f[0] / 2 - 0.1) - (-(f[0] / 2) + 0.1));
        P.pl_min_d2 = (float)(0.1 * 0.1);
    } else {                                // static_defenders.py:221-225,239
        P.pl_xlo = 0.2f; P.pl_xspan = (float)((f[0] / 2 - 0.1) - 0.2);
        P.pl_min_d2 = (float)(0.2 * 0.2);
    }
    P.pl_ylo = (float)(-(f[1] / 2) + 0.1); P.pl_yspan = (float)((f[1] / 2 - 0.1) - (-(f[1] / 2) + 0.1));
    P.ou_theta_dt = (float)(0.17 * M.dt);            // Utils/Utils.py:6,17
    P.ou_sig_sqdt = (float)(0.5 * std::sqrt(M.dt));  // Utils/Utils.py:8,18
    return 0;
}

}  // namespace rsx
