/*
 * selftest.c — drives every entry point of the CPU oracle under AddressSanitizer +
 * UndefinedBehaviorSanitizer (`make -C oracle asan`, run by tests/test_oracle_sanitizers.py).
 * TEST INFRASTRUCTURE like the oracle itself.  It includes the oracle's translation unit, so
 * the sanitizers see the very code the parity tests trust: out-of-bounds indexing of the body /
 * state arrays, signed overflow, invalid float -> int conversions and misaligned accesses abort
 * the run; non-finite states are counted and reported.
 */
#include "rsx_oracle.c"

#include <stdio.h>

static int finite_state(const double* s, int n) {
    for (int i = 0; i < n; ++i) if (!isfinite(s[i])) return 0;
    return 1;
}

static int run_task(int task, int kind, int ft, int nb, int ny, int envs, int steps) {
    int bad = 0;
    for (int i = 0; i < envs; ++i) {
        void* a = rsxo_create_f32(kind, ft, nb, ny, 25);
        void* b = rsxo_create_f64(kind, ft, nb, ny, 25);
        if (!a || !b) return 1000;
        if (rsxo_task_attach_f32(a, task, 77, (uint64_t)i, i % 3 ? 0 : 40)) return 1001;
        if (rsxo_task_attach_f64(b, task, 77, (uint64_t)i, i % 3 ? 0 : 40)) return 1001;
        rsxo_task_reset_f32(a); rsxo_task_reset_f64(b);
        double st[5 + 11 * 22 + RSXO_XROWS], obs[64], rew, info[10];
        uint8_t tm, tr; int n; int64_t m[8];
        for (int t = 0; t < steps; ++t) {
            float act[5] = {(float)((t * 37 + i) % 21 - 10) / 10.0f, (float)((t * 11 + 3 * i) % 21 - 10) / 10.0f,
                            (float)((t * 5) % 21 - 10) / 10.0f, (float)((t % 7) - 3), (float)((t % 5) - 2)};
            rsxo_task_step_f32(a, t % 4 ? NULL : act);
            rsxo_task_step_f64(b, t % 4 ? NULL : act);
            rsxo_task_out_f32(a, obs, &rew, &tm, &tr, info, NULL, &n, m);
            int od = 0;
            (void)od;
            rsxo_get_state_full_f32(a, st);
            bad += !finite_state(st, rsxo_state_dim_f32(a) + RSXO_XROWS) || !isfinite(rew);
            rsxo_get_state_full_f64(b, st);
            bad += !finite_state(st, rsxo_state_dim_f64(b) + RSXO_XROWS);
        }
        rsxo_destroy_f32(a); rsxo_destroy_f64(b);
    }
    return bad;
}

/* the largest configuration: 22 robots in a scrum, every command mode, kicks and chips */
static int run_scrum(int steps) {
    int bad = 0;
    void* e = rsxo_create_f32(1, 1, 11, 11, 25);
    double ball[4] = {0.0, 0.1, 0.0, 0.0}, blue[33], yel[33], st[5 + 11 * 22 + RSXO_XROWS], cm[22 * 8];
    for (int k = 0; k < 11; ++k) {
        blue[3 * k] = 0.2 * (k % 6 - 2.5); blue[3 * k + 1] = 0.2 * (k / 6 - 1.5); blue[3 * k + 2] = 33.0 * k;
        yel[3 * k] = 0.2 * ((k + 11) % 6 - 2.5); yel[3 * k + 1] = 0.2 * ((k + 11) / 6 - 1.5); yel[3 * k + 2] = -47.0 * k;
    }
    rsxo_reset_f32(e, ball, blue, yel);
    for (int t = 0; t < steps; ++t) {
        rsxo_get_state_f32(e, st);
        for (int k = 0; k < 22; ++k) {
            double* q = cm + 8 * k;
            const double* r = st + 5 + 11 * k;
            double gx = st[0] - r[0], gy = st[1] - r[1], n = sqrt(gx * gx + gy * gy) + 1e-9, th = r[2] * RSXO_PI / 180.0;
            memset(q, 0, 8 * sizeof(double));
            if ((t + k) % 9 == 0) { q[0] = 1; q[1] = 60; q[2] = -60; q[3] = 200; q[4] = -200; }   /* wheel-speed mode, saturating */
            else { q[1] = 2 * (gx * cos(th) + gy * sin(th)) / n; q[2] = 2 * (-gx * sin(th) + gy * cos(th)) / n; q[3] = (k % 5) - 2; }
            q[5] = (t + 3 * k) % 11 == 0 ? 4.0 : 0.0; q[6] = (t + k) % 17 == 0 ? 2.5 : 0.0; q[7] = (t + k) % 2;
        }
        rsxo_step_f32(e, cm);
        rsxo_get_state_full_f32(e, st);
        bad += !finite_state(st, rsxo_state_dim_f32(e) + RSXO_XROWS);
    }
    rsxo_destroy_f32(e);
    return bad;
}

/* bodies on top of each other, ball in the middle of a robot, everything at the origin: the contact
 * and reward code divides by distances, and must stay finite where the reference's would not
 * (vss_gym.py:298 divides by |robot - ball| unguarded) */
static int run_coincidence(void) {
    int bad = 0;
    for (int c = 0; c < 6; ++c) {
        const int kind = c % 2, ts = c < 2 ? 25 : 0;   /* time step 0: no physics, the bodies STAY coincident */
        const int task = c < 4 ? (kind ? 2 : 1) : (kind ? 5 : 1);
        void* e = task == 5 ? rsxo_create_f32(1, 2, 2, 0, ts) : rsxo_create_f32(kind, kind ? 2 : 0, kind ? 1 : 3, kind ? 6 : 3, ts);
        rsxo_task_attach_f32(e, task, 1, 0, 0);
        int nb = kind ? 1 : 3, ny = kind ? 6 : 3;
        double ball[4] = {0, 0, 0, 0}, blue[9] = {0}, yel[18] = {0};
        (void)nb; (void)ny;
        rsxo_task_reset_to_f32(e, ball, blue, yel);
        double st[5 + 11 * 22 + RSXO_XROWS], rew;
        for (int t = 0; t < 50; ++t) {
            rsxo_task_step_f32(e, NULL);
            rsxo_task_out_f32(e, NULL, &rew, NULL, NULL, NULL, NULL, NULL, NULL);
            rsxo_get_state_full_f32(e, st);
            bad += !finite_state(st, rsxo_state_dim_f32(e) + RSXO_XROWS) || !isfinite(rew);
        }
        rsxo_destroy_f32(e);
    }
    return bad;
}

int main(void) {
    int bad = 0;
    bad += run_task(1, 0, 0, 3, 3, 24, 400);
    bad += run_task(1, 0, 1, 5, 5, 6, 200);
    bad += run_task(2, 1, 2, 1, 6, 24, 400);
    bad += run_task(3, 1, 2, 1, 4, 12, 400);
    bad += run_task(4, 1, 2, 1, 1, 12, 400);
    bad += run_task(5, 1, 2, 2, 0, 12, 400);
    bad += run_scrum(600);
    bad += run_coincidence();
    uint32_t ctr[4] = {1, 2, 3, 4}, key[2] = {5, 6}, out[4];
    rsxo_philox4x32_10(ctr, key, out); rsxo_philox4x32_7(ctr, key, out);
    printf("oracle selftest: %d non-finite states\n", bad);
    return bad ? 1 : 0;
}
