/* Plain-C caller of the step engine's C-ABI (include/rsx.h): no Python, no torch, no C++.
 * The same calls the reference's rsim.py makes on robosim — construct (rsim.py:116), reset (:38),
 * step (:102), get_state (:105), get_field_params (:50) — then the same pair for a batch through the pinned wire-format buffers,
 * and a short fused VSS-v0 run.
 *
 *   gcc -O2 -Iinclude examples/rsx_c_host.c -o /tmp/rsx_c_host -ldl
 *   /tmp/rsx_c_host rsoccer_amd/librsx_hip.so
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rsx.h"

#define LOAD(name) do { *(void**)(&p_##name) = dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; } } while (0)
#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, p_rsx_last_error()); return 1; } } while (0)

static int (*p_rsx_abi_version)(void);
static const char* (*p_rsx_last_error)(void);
static int (*p_rsx_create)(rsx_sim**, int, int, int, int, int, int, int);
static int (*p_rsx_destroy)(rsx_sim*);
static int (*p_rsx_get_field_params)(const rsx_sim*, double*);
static int (*p_rsx_reset)(rsx_sim*, const double*, const double*, const double*, const uint8_t*, void*);
static int (*p_rsx_step)(rsx_sim*, const double*, void*);
static int (*p_rsx_get_state)(rsx_sim*, double*, void*);
static int (*p_rsx_task_attach)(rsx_sim*, int, uint64_t, uint64_t, int);
static int (*p_rsx_task_reset)(rsx_sim*, void*);
static int (*p_rsx_task_step_n)(rsx_sim*, int, void*);
static int (*p_rsx_read_metrics)(rsx_sim*, int64_t*, void*);
static int (*p_rsx_wire_buffers)(rsx_sim*, double**, double**);
static int (*p_rsx_step_wire)(rsx_sim*, void*);

int main(int argc, char** argv) {
    void* lib = dlopen(argc > 1 ? argv[1] : "rsoccer_amd/librsx_hip.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(rsx_abi_version); LOAD(rsx_last_error); LOAD(rsx_create); LOAD(rsx_destroy); LOAD(rsx_get_field_params);
    LOAD(rsx_reset); LOAD(rsx_step); LOAD(rsx_get_state); LOAD(rsx_task_attach); LOAD(rsx_task_reset);
    LOAD(rsx_task_step_n); LOAD(rsx_read_metrics); LOAD(rsx_wire_buffers); LOAD(rsx_step_wire);
    if (p_rsx_abi_version() != RSX_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }

    /* one VSS 3v3 simulator, driven like robosim.VSS */
    rsx_sim* h = NULL;
    CHECK(p_rsx_create(&h, RSX_KIND_VSS, 0, 3, 3, 25, 1, 0));
    double field[RSX_FIELD_PARAMS];
    CHECK(p_rsx_get_field_params(h, field));
    const double ball[4] = {0.0, 0.0, 0.3, 0.0};
    const double blue[9] = {-0.3, 0.0, 0.0, -0.5, 0.3, 0.0, -0.5, -0.3, 0.0};
    const double yellow[9] = {0.3, 0.0, 180.0, 0.5, 0.3, 180.0, 0.5, -0.3, 180.0};
    CHECK(p_rsx_reset(h, ball, blue, yellow, NULL, NULL));
    double cmds[12] = {20, 20, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   /* blue 0 drives forward */
    double state[41];
    for (int t = 0; t < 40; ++t) CHECK(p_rsx_step(h, cmds, NULL));
    CHECK(p_rsx_get_state(h, state, NULL));
    printf("field %.2f x %.2f m; after 1 s: ball x %.4f, blue0 x %.4f (started at -0.3)\n", field[0], field[1], state[0], state[5]);
    if (!(state[5] > -0.3 + 0.2) || !(state[0] > 0.05)) { fprintf(stderr, "unexpected motion\n"); return 1; }
    CHECK(p_rsx_destroy(h));

    /* 512 simulators in one handle, still in the reference's float64 wire format (ABI 6): the handle owns pinned buffers in that
     * format, the caller fills / reads them in place and the conversion runs on the device — rsim.py:102,105 for a batch */
    {
        enum { B = 512, N = 6, C = 2, S = 5 + 6 * N };
        CHECK(p_rsx_create(&h, RSX_KIND_VSS, 0, 3, 3, 25, B, 0));
        double *wc = NULL, *ws = NULL;
        CHECK(p_rsx_wire_buffers(h, &wc, &ws));
        for (int e = 0; e < B; ++e)
            for (int k = 0; k < N; ++k) { wc[(e * N + k) * C + 0] = k == 0 ? 10.0 + 0.02 * e : 0.0; wc[(e * N + k) * C + 1] = k == 0 ? 10.0 + 0.02 * e : 0.0; }
        for (int t = 0; t < 40; ++t) CHECK(p_rsx_step_wire(h, NULL));
        const double* first = ws;                       /* env e: ws + e * (S + RSX_STATE_EXTRA_ROWS) */
        const double* last = ws + (size_t)(B - 1) * (S + RSX_STATE_EXTRA_ROWS);
        printf("wire buffers, %d envs: blue0 x after 1 s: env 0 %.4f, env %d %.4f (dummy line-up: -0.2; faster wheels go further)\n", B, first[5], B - 1, last[5]);
        if (!(first[5] > -0.2 + 0.1) || !(last[5] > first[5])) { fprintf(stderr, "unexpected motion (wire path)\n"); return 1; }
        CHECK(p_rsx_destroy(h));
    }

    /* 4096 fused VSS-v0 envs, random actions generated on the device */
    CHECK(p_rsx_create(&h, RSX_KIND_VSS, 0, 3, 3, 25, 4096, 0));
    CHECK(p_rsx_task_attach(h, RSX_TASK_VSS_V0, 1234, 0, 0));
    CHECK(p_rsx_task_reset(h, NULL));
    CHECK(p_rsx_task_step_n(h, 2000, NULL));
    int64_t m[RSX_METRICS];
    CHECK(p_rsx_read_metrics(h, m, NULL));
    printf("fused VSS-v0: %lld env-steps, %lld episodes\n", (long long)m[0], (long long)m[1]);
    if (m[0] != 4096LL * 2000) return 1;
    CHECK(p_rsx_destroy(h));
    printf("ok\n");
    return 0;
}
