// rsx_epl.hip — the one-lane-per-env kernels (VSS-v0, SSLStaticDefenders) in their own translation unit:
// built with the compiler's default machine scheduler and explicit occupancy targets per entry point, while
// rsx_api.hip is built with -amdgpu-sched-strategy=max-ilp, which suits the short 8-lanes-per-env kernels at
// small batches but costs these a wave of occupancy.
#include <hip/hip_runtime.h>

#include "rsx_epl.hpp"
#include "rsx_epl_ssl.hpp"

namespace rsx {

void launch_vss_epl(bool rollout, const Params& P, const Buffers& b, int n_steps, hipStream_t s) {
    const int tiles = (P.num_envs + 63) / 64;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8));
    if (rollout)
        hipLaunchKernelGGL(vss_epl_rollout_kernel, grid, dim3(64), 0, s, b.state, b.aux, b.actions, b.flags,
                           P.num_envs, P.state_dim, (int)(grid.x >> 3), n_steps, P, b);
    else
        hipLaunchKernelGGL((vss_epl_kernel<MODE_STEP>), grid, dim3(64), 0, s, b.state, b.aux, b.actions, b.flags,
                           P.num_envs, P.state_dim, (int)(grid.x >> 3), n_steps, P, b);
}

void launch_ssl_sd_epl(bool rollout, const Params& P, const Buffers& b, int n_steps, hipStream_t s) {
    const int tiles = (P.num_envs + 63) / 64;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8));
    if (rollout)
        hipLaunchKernelGGL((ssl_sd_epl_kernel<MODE_ROLLOUT>), grid, dim3(64), 0, s, b.state, b.aux, b.actions, b.flags,
                           P.num_envs, P.state_dim, (int)(grid.x >> 3), n_steps, P, b);
    else
        hipLaunchKernelGGL((ssl_sd_epl_kernel<MODE_STEP>), grid, dim3(64), 0, s, b.state, b.aux, b.actions, b.flags,
                           P.num_envs, P.state_dim, (int)(grid.x >> 3), n_steps, P, b);
}

}  // namespace rsx
