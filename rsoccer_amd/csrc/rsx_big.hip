// rsx_big.hip — the lane-group kernels once more, compiled for LARGE batches (own translation unit, own flags).
//
// rsx_api.hip builds task_step_kernel with -amdgpu-sched-strategy=max-ilp: at the benchmark batches a SIMD holds one
// wave and latency has to be hidden inside it.  From a few thousand waves on the opposite holds — registers, i.e.
// waves per SIMD, are what hides latency — so the configurations that have no one-lane-per-env kernel (SSL 11v11:
// 32 lanes per env) are built here a second time with the default scheduler and without the SLP vectorizer, under
// another symbol name, and the host picks by batch size (RSX_BIG_MIN_ENVS).  Same source, same results.
#include <hip/hip_runtime.h>

#include "rsx_launch.hpp"

#define task_step_kernel task_step_kernel_big
#include "rsx_kernels.hpp"

namespace rsx {

void launch_scrimmage_big(bool rollout, const Params& P, const Buffers& b, int n_steps, hipStream_t s) {
    constexpr int L = 32, G = 64 / L;
    const int tiles = (P.num_envs + G - 1) / G;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 8));
    if (rollout)
        rsx_launch((task_step_kernel_big<RSX_KIND_SSL, 32, RSX_TASK_SSL_SCRIMMAGE, 22, MODE_ROLLOUT>), grid, dim3(64), 0, s, b.state, b.aux,
                           b.actions, b.flags, P.num_envs, RSX_HOT_DIM(P.state_dim, P.row_stride, P.num_envs), (int)(grid.x >> 3), n_steps, P, b);
    else
        rsx_launch((task_step_kernel_big<RSX_KIND_SSL, 32, RSX_TASK_SSL_SCRIMMAGE, 22, MODE_STEP>), grid, dim3(64), 0, s, b.state, b.aux,
                           b.actions, b.flags, P.num_envs, RSX_HOT_DIM(P.state_dim, P.row_stride, P.num_envs), (int)(grid.x >> 3), n_steps, P, b);
}

}  // namespace rsx
