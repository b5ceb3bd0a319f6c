/* VSS scrum on the CPU oracle: every robot drives at the ball (differential drive, proportional heading control), the deepest
 * robot - robot overlap of every env is sampled every step (JAM_VERBOSE=1 lists every sample above 1.2 cm with the two robots' positions).
 * Round 6 used it with temporary experiment knobs in the oracle to pick model v2 of the VSS class (profiles/r06_jam_vss_model_v2.txt).
 *   gcc -O2 -fopenmp -ffp-contract=off -mfma -o /tmp/exp_jam_vss tools/exp_jam_vss.c -lm && /tmp/exp_jam_vss [ft] [nb] [envs] [steps] */
#include "../oracle/rsx_oracle.c"
#include <stdio.h>
static int cmpf(const void* a, const void* b) { float x = *(const float*)a, y = *(const float*)b; return x < y ? -1 : x > y; }
int main(int argc, char** argv) {
    int ft = argc > 1 ? atoi(argv[1]) : 0, nb = argc > 2 ? atoi(argv[2]) : 3, B = argc > 3 ? atoi(argv[3]) : 256, T = argc > 4 ? atoi(argv[4]) : 1000;
    int N = 2 * nb;
    float* samp = malloc(sizeof(float) * (size_t)B * T);
    double worst_all = 0;
#pragma omp parallel for schedule(dynamic)
    for (int e = 0; e < B; ++e) {
        void* s = rsxo_create_f32(0, ft, nb, nb, 25);
        uint32_t key[2] = {12345u, 99u};
        double blue[3 * 11], yel[3 * 11], ball[4] = {0, 0, 0, 0};
        double fl = ft == 0 ? 1.5 : 2.2, fw = ft == 0 ? 1.3 : 1.8;
        for (int k = 0; k < N; ++k) {   /* jittered grid */
            uint32_t ctr[4] = {(uint32_t)e, (uint32_t)k, 0, 7}, u[4]; rsxo_philox4x32_7(ctr, key, u);
            double* p = k < nb ? blue + 3 * k : yel + 3 * (k - nb);
            int cols = (N + 1) / 2;
            p[0] = (k % cols - (cols - 1) / 2.0) * (fl * 0.7 / cols) + (u[0] / 4294967296.0 - 0.5) * 0.02;
            p[1] = (k / cols ? 0.3 : -0.3) + (u[1] / 4294967296.0 - 0.5) * 0.02;
            p[2] = u[2] / 4294967296.0 * 360.0 - 180.0;
        }
        { uint32_t ctr[4] = {(uint32_t)e, 99, 0, 7}, u[4]; rsxo_philox4x32_7(ctr, key, u);
          ball[0] = (u[0] / 4294967296.0 - 0.5) * fl * 0.8; ball[1] = (u[1] / 4294967296.0 - 0.5) * fw * 0.3; }
        rsxo_reset_f32(s, ball, blue, yel);
        double st[5 + 6 * 22 + 2], cm[2 * 22];
        for (int t = 0; t < T; ++t) {
            rsxo_get_state_f32(s, st);
            for (int k = 0; k < N; ++k) {
                const double* r = st + 5 + 6 * k;
                double ang = atan2(st[1] - r[1], st[0] - r[0]), th = r[2] * RSXO_PI / 180.0;
                double err = ang - th; while (err > RSXO_PI) err -= 2 * RSXO_PI; while (err < -RSXO_PI) err += 2 * RSXO_PI;
                uint32_t ctr[4] = {(uint32_t)e, (uint32_t)k, (uint32_t)t + 1, 8}, u[4]; rsxo_philox4x32_7(ctr, key, u);
                double v = 0.9 * (cos(err) > 0 ? cos(err) : 0.0) + 0.1, w = 8.0 * err + (u[0] / 4294967296.0 - 0.5) * 4.0;
                cm[2 * k] = (v - w * 0.04) / 0.026; cm[2 * k + 1] = (v + w * 0.04) / 0.026;
            }
            rsxo_step_f32(s, cm);
            rsxo_get_state_f32(s, st);
            double dmin = 9;
            for (int i = 0; i < N; ++i) for (int j = i + 1; j < N; ++j) {
                double dx = st[5 + 6 * i] - st[5 + 6 * j], dy = st[6 + 6 * i] - st[6 + 6 * j], d = sqrt(dx * dx + dy * dy);
                if (d < dmin) dmin = d;
            }
            double ov = 0.075 - dmin; if (ov < 0) ov = 0;
            if (ov > 0.012 && getenv("JAM_VERBOSE")) {
                for (int i = 0; i < N; ++i) for (int j = i + 1; j < N; ++j) {
                    double dx = st[5 + 6 * i] - st[5 + 6 * j], dy = st[6 + 6 * i] - st[6 + 6 * j], d = sqrt(dx * dx + dy * dy);
                    if (d == dmin) printf("env %d t %d ov %.2f cm: robots %d (%.4f, %.4f) %d (%.4f, %.4f)\n", e, t, 100 * ov, i, st[5 + 6 * i], st[6 + 6 * i], j, st[5 + 6 * j], st[6 + 6 * j]);
                }
            }
            samp[(size_t)e * T + t] = (float)ov;
        }
        rsxo_destroy_f32(s);
    }
    size_t n = (size_t)B * T, nz = 0;
    for (size_t i = 0; i < n; ++i) { if (samp[i] > worst_all) worst_all = samp[i]; nz += samp[i] > 0; }
    qsort(samp, n, sizeof(float), cmpf);
    printf("ft %d %dv%d envs %d steps %d: touching samples %.1f %%, worst %.2f cm, p99.9 %.2f, p99 %.2f cm, p90 %.2f, median %.2f\n", ft, nb, nb, B, T,
           100.0 * nz / n, 100 * worst_all, 100 * samp[(size_t)(0.999 * n)], 100 * samp[(size_t)(0.99 * n)], 100 * samp[(size_t)(0.9 * n)], 100 * samp[n / 2]);
    return 0;
}
