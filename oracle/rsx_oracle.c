/*
 * rsx_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * nothing under rsoccer_amd/ links, imports or calls it.
 *
 * What it restates
 *   (1) PHYSICS — the per-env.step() 2-D rigid-body update that, in the reference, happens
 *       inside the third-party module `robosim` (PyPI rc-robosim >= 1.2.0, pinned at
 *       /root/reference/setup.py:15; call sites rsoccer_gym/Simulators/rsim.py:38,50,102,105,
 *       116,155,158,169).  That module's source is NOT in /root/reference, is not installed and
 *       cannot be fetched, and the reference holds no test, golden vector or fixture for it
 *       (its only test is Utils/kdtree_test.py).  ==> PHYSICS PARITY WITH rSim IS UNPINNED.
 *       The model here is the build's own 2-D specification (DESIGN.md "Physics model"); this
 *       file is its executable definition and the HIP kernels are checked against it
 *       (bit-exact for the f32 instantiation).
 *   (2) TASK ARITHMETIC — observation / command / reward / done / placement / OU-noise
 *       formulas of VSS-v0 (rsoccer_gym/vss/env_vss/vss_gym.py:93-311),
 *       SSLStaticDefenders-v0 (rsoccer_gym/ssl/ssl_hw_challenge/static_defenders.py:90-322),
 *       SSLDribbling-v0 (dribbling.py:76-202), SSLContestedPossession-v0
 *       (contested_possession.py:78-227) and SSLPassEndurance-v0 (pass_endurance.py:77-233).
 *       These ARE pinned: tests/test_oracle_golden.py checks them against vectors captured by
 *       importing the reference (tests/golden/make_golden.py).
 *
 * The file is compiled once and instantiates every routine twice (float: the exact model the
 * GPU implements; double: the reference's own boundary precision) via rsx_oracle_impl.h.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no fast-math — required for the
 * bit-exact comparison).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Field tables and model constants (double).  Provenance: "ref" = literal visible in
 * /root/reference, "build" = chosen by this project (the values only
 * exist inside rc-robosim, which is absent).
 * ---------------------------------------------------------------------------------------- */
typedef struct rsxo_cfg {
    int kind, field_type, n_blue, n_yellow, n_robots, n_bodies, time_step_ms, n_sub;
    /* Field.py:5-21 order */
    double field[17];
    /* derived geometry */
    double half_len, half_wid, goal_half_wid, goal_depth, margin;
    double r_robot, r_ball;
    /* dynamics */
    double h;                 /* sub-step seconds */
    double m_robot, m_ball;
    double a_lin, a_lat, a_ang, mu_g;
    double e_rr, e_rb, e_wall_ball, e_wall_robot, beta;
    double w_max, r_wheel, lever;
    double grav, e_ground, vz_min, robot_h;
    double dck, half_kw, ir_tol, drib_vmax;
    double mu_rr, mu_rb, mu_wb, spin_dec;   /* Coulomb friction in contacts, spin deceleration (rad/s^2) */
    double pen2;                            /* overlap beyond which an env gets the second contact sweep */
    double wheel_ang[4];
    double pinv[3][4];
} rsxo_cfg;

#define RSXO_PI 3.14159265358979323846
#define RSXO_XROWS 2 /* internal state entries behind get_state(): ball vz, ball spin */

static int rsxo_cfg_init(rsxo_cfg* c, int kind, int field_type, int nb, int ny, int ts_ms) {
    memset(c, 0, sizeof(*c));
    if (kind != 0 && kind != 1) return -1;
    if (nb < 0 || ny < 0 || nb + ny < 1 || nb + ny > 22 || ts_ms < 0) return -1;
    c->kind = kind; c->field_type = field_type; c->n_blue = nb; c->n_yellow = ny;
    c->n_robots = nb + ny; c->n_bodies = nb + ny + 1; c->time_step_ms = ts_ms;
    /* sub-stepping: 5 ms sub-steps (build) */
    c->n_sub = (ts_ms + 4) / 5;
    c->h = c->n_sub ? (ts_ms * 0.001) / c->n_sub : 0.0;
    double* f = c->field;
    if (kind == 0) {
        if (field_type == 0) {        /* ref: Render/field.py:190-199 */
            f[0] = 1.5; f[1] = 1.3; f[2] = 0.15; f[3] = 0.7; f[4] = 0.4; f[5] = 0.1;
        } else if (field_type == 1) { /* build: 5v5 */
            f[0] = 2.2; f[1] = 1.8; f[2] = 0.15; f[3] = 0.8; f[4] = 0.4; f[5] = 0.15;
        } else return -1;
        f[6] = 0.0215;                /* ref: Render/ball.py:6 */
        f[7] = 0.0; f[8] = 0.0; f[9] = 0.0;
        f[10] = 90.0; f[11] = 270.0; f[12] = 0.0; f[13] = 0.0;
        f[14] = 0.0375;               /* ref: vss_gym_base.py:57 */
        f[15] = 0.026; f[16] = 440.0; /* build */
        c->m_robot = 0.18; c->m_ball = 0.046;
        c->a_lin = 8.0; c->a_lat = 20.0; c->a_ang = 300.0; c->mu_g = 0.3;
        c->e_rr = 0.1; c->e_rb = 0.3; c->e_wall_ball = 0.6; c->e_wall_robot = 0.1;
        c->margin = 0.0;
        c->lever = 0.04;              /* ref: vss_gym_base.py:58 */
    } else {
        if (field_type == 0) {        /* ref: Render/field.py:253-262 */
            f[0] = 9.0; f[1] = 6.0; f[2] = 1.0; f[3] = 2.0; f[4] = 1.0; f[5] = 0.18;
        } else if (field_type == 1) { /* build: division A */
            f[0] = 12.0; f[1] = 9.0; f[2] = 1.8; f[3] = 3.6; f[4] = 1.8; f[5] = 0.18;
        } else if (field_type == 2) { /* build: hardware-challenge field */
            f[0] = 6.0; f[1] = 4.0; f[2] = 0.8; f[3] = 2.0; f[4] = 1.0; f[5] = 0.18;
        } else return -1;
        f[6] = 0.0215;
        f[7] = 0.073; f[8] = 0.005; f[9] = 0.08;          /* build */
        f[10] = 60.0; f[11] = 135.0; f[12] = 225.0; f[13] = 300.0; /* build */
        f[14] = 0.09;                 /* ref: ssl_gym_base.py:58 */
        f[15] = 0.02475;              /* build */
        f[16] = 160.0 * 60.0 / (2.0 * RSXO_PI); /* 160 rad/s, ref: static_defenders.py:71 */
        c->m_robot = 2.2; c->m_ball = 0.046;
        c->a_lin = 5.0; c->a_lat = 0.0; c->a_ang = 50.0; c->mu_g = 0.4;
        c->e_rr = 0.1; c->e_rb = 0.2; c->e_wall_ball = 0.5; c->e_wall_robot = 0.1;
        c->margin = 0.3;
        c->lever = 0.09;
    }
    c->beta = 0.8;
    c->half_len = f[0] / 2; c->half_wid = f[1] / 2;
    c->goal_half_wid = f[4] / 2; c->goal_depth = f[5];
    c->r_ball = f[6]; c->r_robot = f[14];
    c->r_wheel = f[15];
    c->w_max = f[16] / 60.0 * 2.0 * RSXO_PI;
    c->grav = 9.81; c->e_ground = 0.5; c->vz_min = 0.2; c->robot_h = 0.15;
    c->dck = f[7]; c->half_kw = f[9] / 2; c->ir_tol = 0.025; c->drib_vmax = 1.0;
    c->mu_rr = 0.2; c->mu_rb = 0.35; c->mu_wb = 0.3; c->spin_dec = 30.0;   /* build */
    c->pen2 = 0.005;                                                       /* build */
    if (getenv("RSXO_PEN2")) c->pen2 = atof(getenv("RSXO_PEN2"));          /* model experiments only */
    for (int k = 0; k < 4; ++k) c->wheel_ang[k] = f[10 + k] * RSXO_PI / 180.0;
    if (kind == 1) {
        /* omni inverse kinematics: wheel surface speed_k = -sin(a_k) vx + cos(a_k) vy + R w.
         * pinv = (J^T J)^-1 J^T, J = 4x3. */
        double J[4][3], A[3][3] = {{0}}, Ai[3][3];
        for (int k = 0; k < 4; ++k) {
            J[k][0] = -sin(c->wheel_ang[k]); J[k][1] = cos(c->wheel_ang[k]); J[k][2] = c->r_robot;
        }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 4; ++k) A[i][j] += J[k][i] * J[k][j];
        double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1])
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
            double s = 0; for (int j = 0; j < 3; ++j) s += Ai[i][j] * J[k][j];
            c->pinv[i][k] = s;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Philox4x32 (Salmon et al., SC'11) — counter-based RNG shared by placement, OU noise and
 * random actions.  counter = (global env id, episode, tick, domain), key = (seed lo, seed hi).
 * ---------------------------------------------------------------------------------------- */
void rsxo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* the published 10-round form (Random123 known-answer vectors pin the round function) */
void rsxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { rsxo_philox4x32(ctr, key, 10, out); }
/* what the engine draws with: Philox4x32-7, the smallest round count of the family that passes
 * BigCrush (Salmon et al., SC'11, table 2); 32-bit multiplies are quarter rate on CDNA */
#define RSXO_PHILOX_ROUNDS 7
void rsxo_philox4x32_7(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { rsxo_philox4x32(ctr, key, RSXO_PHILOX_ROUNDS, out); }

/* domains.  ACT: per-step block(s) of an env: block q in bits 8.. ; SSL tasks use block 0 for the
 * agent's action; VSS-v0 gives robot k the words (2 (k & 1), 2 (k & 1) + 1) of block k >> 1
 * (robot 0: random action, robots >= 1: the two uniforms of their Box-Muller OU draw) */
#define RSXO_DOM_ACT   1u
#define RSXO_DOM_PLACE 3u
#define RSXO_DOM_RAW   4u /* rsx_step_dev_random: counter = (env id, tick, robot, RAW) */

/* float-only elementary functions with a fixed operation order (mirrored instruction for
 * instruction by rsoccer_amd/csrc/rsx_math.hpp); coefficients: Cephes sinf/cosf/logf.
 * Fused multiply-adds appear ONLY where written (fmaf: one rounding, identical on x86 -mfma,
 * glibc's software fmaf and the GPU's v_fma_f32); the build keeps -ffp-contract=off. */
static inline void rsxo_sincos_f32(float a, float* s, float* c) {
    float t = a * 0.636619772f;
    int k = (int)(t + (t >= 0.0f ? 0.5f : -0.5f));
    float fk = (float)k;
    float r = fmaf(fk, -7.54978995489188e-8f, fmaf(fk, -4.837512969970703125e-4f, fmaf(fk, -1.5703125f, a)));
    float z = r * r;
    float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                    fmaf(-0.5f, z, 1.0f));
    switch (k & 3) {
        case 0: *s = ps; *c = pc; break;
        case 1: *s = pc; *c = -ps; break;
        case 2: *s = -ps; *c = -pc; break;
        default: *s = -pc; *c = ps; break;
    }
}
static inline float rsxo_log_f32(float x) { /* x in [2^-24, 1] */
    uint32_t ix; memcpy(&ix, &x, 4);
    int e = (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) | 0x3f800000u;
    float m; memcpy(&m, &ix, 4);
    if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }
    float f = m - 1.0f, z = f * f;
    float p = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(7.0376836292e-2f, f, -1.1514610310e-1f), f, 1.1676998740e-1f), f,
                    -1.2420140846e-1f), f, 1.4249322787e-1f), f, -1.6668057665e-1f), f, 2.0000714765e-1f), f,
                    -2.4999993993e-1f), f, 3.3333331174e-1f) * f * z;
    float fe = (float)e;
    p = fmaf(fe, -2.12194440e-4f, p);
    p = fmaf(-0.5f, z, p);
    float r = f + p;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

#ifdef _OPENMP
#include <omp.h>
void rsxo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
void rsxo_set_threads(int n) { (void)n; }
#endif

/* atan2 for the float instantiation (Cephes atanf; only used by the PassEndurance placement) */
static inline float rsxo_atan_f32(float x) {
    float sgn = x < 0.0f ? -1.0f : 1.0f;
    x = fabsf(x);
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return sgn * y;
}
static inline float rsxo_atan2_f32(float y, float x) {
    if (x > 0.0f) return rsxo_atan_f32(y / x);
    if (x < 0.0f) return rsxo_atan_f32(y / x) + (y >= 0.0f ? 3.14159265358979f : -3.14159265358979f);
    return y > 0.0f ? 1.5707963267948966f : (y < 0.0f ? -1.5707963267948966f : 0.0f);
}

/* ---- instantiate: float ---- */
#define R float
#define R_ATAN2(y, x) rsxo_atan2_f32((y), (x))
#define SUF(n) n##_f32
#define R_SINCOS(a, s, c) rsxo_sincos_f32((a), (s), (c))
#define R_LOG(x) rsxo_log_f32(x)
#define R_SQRT(x) sqrtf(x)
#define R_FABS(x) fabsf(x)
#define R_FMA(a, b, c) fmaf((a), (b), (c))
#include "rsx_oracle_impl.h"
#undef R
#undef SUF
#undef R_SINCOS
#undef R_LOG
#undef R_SQRT
#undef R_FABS
#undef R_FMA
#undef R_ATAN2

/* ---- instantiate: double ---- */
static inline void rsxo_sincos_f64(double a, double* s, double* c) { *s = sin(a); *c = cos(a); }
#define R double
#define SUF(n) n##_f64
#define R_SINCOS(a, s, c) rsxo_sincos_f64((a), (s), (c))
#define R_LOG(x) log(x)
#define R_SQRT(x) sqrt(x)
#define R_FABS(x) fabs(x)
#define R_FMA(a, b, c) fma((a), (b), (c))
#define R_ATAN2(y, x) atan2((y), (x))
#include "rsx_oracle_impl.h"
