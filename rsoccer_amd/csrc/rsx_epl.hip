// rsx_epl.hip — the one-lane-per-env kernels (VSS-v0 and the four SSL tasks) in their own translation unit:
// built with the compiler's default machine scheduler and explicit occupancy targets per entry point, while
// rsx_api.hip is built with -amdgpu-sched-strategy=max-ilp, which suits the short 8-lanes-per-env kernels at
// small batches but costs these a wave of occupancy.
#include <hip/hip_runtime.h>

#include "rsx_launch.hpp"

#include <cstdlib>

#include "rsx_epl.hpp"
#include "rsx_epl_ssl.hpp"
#include "rsx_quad_ssl.hpp"

namespace rsx {

int ssl_quad_grid(int num_envs);
int epl_grid(int num_envs);

// development knob: RSX_EPL_LDS_PAD=<bytes> of dynamic LDS per workgroup limit the waves per CU (occupancy
// sensitivity measurements); 0 in production
// tile order of a single-step launch: alternates with the step counter (tile_of_block_zigzag, rsx_kernels.hpp);
// RSX_EPL_ZIGZAG=0 keeps one direction (development A/B)
// (device-keyed launches — n_steps carries RSX_TICK_DEV — cannot be told the parity: a negative `per` asks the kernel to alternate)
static int step_per_xcd(const Params& P, const dim3& grid, const int n_steps) {
    static const bool zig = !(std::getenv("RSX_EPL_ZIGZAG") && std::atoi(std::getenv("RSX_EPL_ZIGZAG")) == 0);
    const int per = (int)(grid.x >> 3);
    if (n_steps & RSX_TICK_DEV) return zig ? -per : per;
    return (zig && (P.tick_base & 1u)) ? -per : per;
}

#ifndef RSX_SD_LEAN_MAX_ENVS
#define RSX_SD_LEAN_MAX_ENVS 786432
#endif

static unsigned epl_lds_pad() {
    static const unsigned pad = std::getenv("RSX_EPL_LDS_PAD") ? (unsigned)std::atoi(std::getenv("RSX_EPL_LDS_PAD")) : 0u;
    return pad;
}

void launch_vss_epl(bool rollout, const Params& P, const Buffers& b, int n_steps, hipStream_t s) {
    const int tiles = (P.num_envs + 63) / 64;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8));
    if (rollout)
        rsx_launch(vss_epl_rollout_kernel, grid, dim3(64), 0, s, b.state, b.aux, b.actions, b.flags,
                           P.num_envs, RSX_HOT_DIM(P.state_dim, P.row_stride, P.num_envs), (int)(grid.x >> 3), n_steps, P, b);
    else
        rsx_launch((vss_epl_kernel<MODE_STEP>), grid, dim3(64), epl_lds_pad(), s, b.state, b.aux, b.actions, b.flags,
                           P.num_envs, RSX_HOT_DIM(P.state_dim, P.row_stride, P.num_envs), step_per_xcd(P, grid, n_steps), n_steps, P, b);
}

template <int TASK>
static void launch_ssl_epl_t(bool rollout, const Params& P, const Buffers& b, int n_steps, hipStream_t s) {
    const int tiles = (P.num_envs + 63) / 64;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8));
    if (rollout)
        rsx_launch((ssl_epl_kernel<TASK, MODE_ROLLOUT>), grid, dim3(64), 0, s, b.state, b.aux, b.actions, b.flags,
                           P.num_envs, RSX_HOT_DIM(P.state_dim, P.row_stride, P.num_envs), (int)(grid.x >> 3), n_steps, P, b);
    else {
        // the lean single-step form (rsx_epl_ssl.hpp): 1v6 below RSX_SD_LEAN_MAX_ENVS (occupancy-bound there: 262 144 envs 52 -> 47 us;
        // at 1 M envs, bandwidth-bound, the classic form is 4-7 % faster), contested possession always (-5 %); measured equal
        // or worse for dribbling and pass endurance (episodes of a few steps: most waves store the robot rows twice)
        const char* const fe = std::getenv("RSX_EPL_LEAN");   // 0 / 1: tests and A/B runs (read per launch: tests switch it inside one process)
        const int force = fe ? std::atoi(fe) : -1;
        const bool lean = force >= 0 ? force != 0
                                     : TASK == RSX_TASK_SSL_CONTESTED || (TASK == RSX_TASK_SSL_STATIC_DEFENDERS && P.num_envs < RSX_SD_LEAN_MAX_ENVS);
        if (lean && (TASK == RSX_TASK_SSL_CONTESTED || TASK == RSX_TASK_SSL_STATIC_DEFENDERS))
            rsx_launch((ssl_epl_kernel<TASK, MODE_STEP, (TASK == RSX_TASK_SSL_CONTESTED || TASK == RSX_TASK_SSL_STATIC_DEFENDERS)>), grid, dim3(64), 0, s,
                               b.state, b.aux, b.actions, b.flags, P.num_envs, RSX_HOT_DIM(P.state_dim, P.row_stride, P.num_envs), step_per_xcd(P, grid, n_steps), n_steps, P, b);
        else
            rsx_launch((ssl_epl_kernel<TASK, MODE_STEP>), grid, dim3(64), 0, s, b.state, b.aux, b.actions, b.flags,
                               P.num_envs, RSX_HOT_DIM(P.state_dim, P.row_stride, P.num_envs), step_per_xcd(P, grid, n_steps), n_steps, P, b);
    }
}

void launch_ssl_quad(const Params& P, const Buffers& b, int n_steps, hipStream_t s) {   // SSL 11v11 scrimmage, four lanes per env, single-step launches (n_steps = 1 | flags)
    const dim3 grid((unsigned)ssl_quad_grid(P.num_envs));
    rsx_launch((ssl_quad_kernel<MODE_STEP>), grid, dim3(64), 0, s, b.state, b.aux, b.actions, b.flags,
                       P.num_envs, RSX_HOT_DIM(P.state_dim, P.row_stride, P.num_envs), step_per_xcd(P, grid, n_steps), n_steps, P, b);
}

// workgroups of a launch (the host sizes the per-workgroup tick slots from these: rsx_kernels.hpp, step_tick)
int ssl_quad_grid(int num_envs) { const int tiles = (num_envs + Q_ENVS - 1) / Q_ENVS; return ((tiles + 7) / 8) * 8; }
int epl_grid(int num_envs) { const int tiles = (num_envs + 63) / 64; return ((tiles + 7) / 8) * 8; }

void launch_ssl_epl(int task, bool rollout, const Params& P, const Buffers& b, int n_steps, hipStream_t s) {
    switch (task) {
        case RSX_TASK_SSL_STATIC_DEFENDERS: launch_ssl_epl_t<RSX_TASK_SSL_STATIC_DEFENDERS>(rollout, P, b, n_steps, s); break;
        case RSX_TASK_SSL_DRIBBLING: launch_ssl_epl_t<RSX_TASK_SSL_DRIBBLING>(rollout, P, b, n_steps, s); break;
        case RSX_TASK_SSL_CONTESTED: launch_ssl_epl_t<RSX_TASK_SSL_CONTESTED>(rollout, P, b, n_steps, s); break;
        default: launch_ssl_epl_t<RSX_TASK_SSL_PASS_ENDURANCE>(rollout, P, b, n_steps, s); break;
    }
}

}  // namespace rsx

#ifdef RSX_QSTATS
extern "C" __attribute__((visibility("default"))) int rsx_debug_qstats(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rsx::rsx_qstats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rsx::rsx_qstats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
