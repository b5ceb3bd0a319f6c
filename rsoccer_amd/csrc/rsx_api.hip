// rsx_api.hip — host side of librsx_hip.so: the C-ABI declared in include/rsx.h.
//
// Stands where the pybind11 module `robosim` stands in the reference (constructed at
// rsoccer_gym/Simulators/rsim.py:116-124,169-177; stepped at :102,:155; read at :105,:158;
// reset at :38; field at :50; destroyed at :41).  HIP only: there is no CPU path in this library.
#include <hip/hip_runtime.h>

#include "rsx_launch.hpp"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "rsx.h"
#include "rsx_kernels.hpp"

namespace rsx {
// rsx_epl.hip (own translation unit, own compiler flags)
void launch_vss_epl(bool rollout, const Params& P, const Buffers& b, int n_steps, hipStream_t s);
void launch_ssl_epl(int task, bool rollout, const Params& P, const Buffers& b, int n_steps, hipStream_t s);
// rsx_big.hip: the 32-lanes-per-env kernel of the SSL 11v11 scrimmage task built for large batches
void launch_scrimmage_big(bool rollout, const Params& P, const Buffers& b, int n_steps, hipStream_t s);
void launch_ssl_quad(const Params& P, const Buffers& b, int n_steps, hipStream_t s);   // rsx_quad_ssl.hpp: four lanes per env, single-step launches
int ssl_quad_grid(int num_envs);   // workgroups of those launches
int epl_grid(int num_envs);
}

using namespace rsx;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(RSX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

}  // namespace

// smallest batch stepped by the one-lane-per-env kernels: measured crossovers of single-step launches (round 3,
// gpurun_out/xover.txt -> profiles/r03_layout_crossovers.txt; multi-step launches cross earlier): VSS-v0 98 304 envs
// (31.5 vs 29.3 us), static defenders 65 536 (29.2 vs 28.5), dribbling 49 152 (30.0 vs 29.2), contested possession
// 32 768 (19.5 vs 18.8), pass endurance 32 768 (17.8 vs 15.3)
#ifndef RSX_EPL_MIN_ENVS
#define RSX_EPL_MIN_ENVS 98304
#endif
// smallest batch of the SSL 11v11 scrimmage task stepped by the large-batch build of its kernel (rsx_big.hip)
#ifndef RSX_BIG_MIN_ENVS
#define RSX_BIG_MIN_ENVS 8192
#endif
// smallest batches of the scrimmage task stepped by the four-lanes-per-env kernel (RSX_LAYOUT=quad|lanes overrides)
#ifndef RSX_QUAD_MIN_ENVS
#define RSX_QUAD_MIN_ENVS 32768
#endif
#ifndef RSX_QUAD_MIN_ENVS_CROWDED
#define RSX_QUAD_MIN_ENVS_CROWDED 65536
#endif
// from this batch on a multi-step call (rsx_task_rollout) on a four-lane handle is issued as single-step launches
#ifndef RSX_QUAD_ROLLOUT_MIN_ENVS
#define RSX_QUAD_ROLLOUT_MIN_ENVS 49152
#endif
#ifndef RSX_QUAD_ROLLOUT_MIN_ENVS_CROWDED
#define RSX_QUAD_ROLLOUT_MIN_ENVS_CROWDED 196608
#endif
#ifndef RSX_EPL_MIN_ENVS_SSL
#define RSX_EPL_MIN_ENVS_SSL 65536
#endif
// largest batch whose single-step launches carry placement-helper workgroups (rsx_kernels.hpp: placement_helper): where a
// launch is as long as its slowest wave and half of the SIMDs are idle anyway
#ifndef RSX_PCACHE_MAX_ENVS
#define RSX_PCACHE_MAX_ENVS 16384
#endif
// largest batch whose host-format step (rsx_step / rsx_step_state) lets the kernel read the commands from, and mirror the
// state into, pinned host memory (one launch + one synchronisation; PCIe latency instead of two copy engines' worth of it)
#ifndef RSX_ZERO_COPY_MAX_ENVS
#define RSX_ZERO_COPY_MAX_ENVS 64
#endif

struct rsx_sim {
    Params P;
    HostModel M;
    int device = 0;
    int field_type = 0, time_step_ms = 0;   // as given to rsx_create (checkpoint header)
    int L = 8;   // lanes per env
    int NR = 0;  // compile-time robot count of the selected kernel variant (0 = generic)
    bool quad = false; // SSL 11v11 scrimmage: single-step launches of a large batch use the four-lanes-per-env kernel (rsx_quad_ssl.hpp)
    bool big = false;  // SSL 11v11 scrimmage: step / rollout launches use the large-batch build of the 32-lane kernel (rsx_big.hip)
    bool epl = false;  // VSS-v0 3v3 / the registered SSL tasks: step and rollout launches use the one-lane-per-env kernels (large batches)
    // one allocation per lifetime stage (few pages -> few TLB entries per launch)
    char* arena_sim = nullptr;   // state | cmds
    char* arena_task = nullptr;  // aux | obs | final_obs | flags | actions | metrics
    float* d_state = nullptr;
    float* d_state_alt = nullptr;   // second state buffer of rsx_step_dev_flip (allocated on first use)
    float* alt_alloc = nullptr;     // ... the allocation behind it: d_state and d_state_alt trade places at every flip, this is what is freed
    float* d_cmds = nullptr;
    float *d_aux = nullptr, *d_obs = nullptr, *d_final_obs = nullptr, *d_actions = nullptr;
    uint8_t* d_flags = nullptr;
    unsigned long long* d_metrics = nullptr;
    unsigned long long* d_mslots = nullptr;   // [MSLOTS][RSX_METRICS] partial episode counters (metric_slot)
    float* d_pcache = nullptr;                // placement cache of the latency-bound batches (rsx_kernels.hpp: placement_helper), or null
    unsigned long long* d_pcstats = nullptr;  // [2] cache hits / inline placements (RSX_PCACHE_STATS=1)
    unsigned long long* d_check = nullptr;   // rsx_check_finite counter
    std::vector<float> h_f32;
    // host-format path: pinned staging; rsx_step() brings the new state back with its own
    // synchronisation, so the rsx_get_state() that follows it (rsim.py:102 then :105) is a pure
    // host conversion.  The copy is trusted only while every state change went through this API:
    // handing out raw device pointers (rsx_dev_view_get) switches the shortcut off for good.
    float* pin_cmds = nullptr;
    float* pin_state = nullptr;
    float* pin_cmds_dev = nullptr;            // the same two buffers as the device sees them (zero-copy path of small batches)
    float* pin_state_dev = nullptr;
    // batches above RSX_ZERO_COPY_MAX_ENVS: the reference's wire format itself (float64, [B][N*C] commands, [B][state_dim + 2] state) in
    // pinned host memory; small kernels convert between it and the f32 SoA arrays ON THE DEVICE, reading / writing the pinned buffers
    // across PCIe — no transposing loop on a CPU thread, no staging copy (rsx_wire_buffers / rsx_step_wire; rsx_step / rsx_get_state
    // are a memcpy in front of / behind them)
    double* wire_cmds = nullptr;
    double* wire_state = nullptr;
    double* wire_cmds_dev = nullptr;
    double* wire_state_dev = nullptr;
    bool host_state_valid = false;
    bool host_state_cache = true;
    bool task_ready = false;   // a reset has opened the first episode
    size_t arena_task_bytes = 0, pcache_bytes = 0;   // sizes of arena_task and of the placement cache inside it (rsx_task_reseed re-initialises them)
    uint32_t tick = 0;                        // fused steps taken since attach (key of the per-step draws); stale once tick_dev is set
    // rsx_task_enable_capture: the step counter lives in device memory (one slot per workgroup behind the metrics vector,
    // rsx_kernels.hpp: step_tick) so that captured stepping launches advance it when a graph replays them
    bool tick_dev = false;
    int tick_slots = 0;                       // workgroups of the handle's per-step launches: the slots every stepping call keeps in sync
    int tick_slots_alloc = 0;                 // slots allocated (the largest grid any layout of this batch could launch): rsx_task_enable_capture and
                                              // rsx_task_checkpoint_load write ALL of them, so that no grid ever reads a slot nobody has set
};

namespace {

int pick_lanes(int n_bodies) {
    int L = n_bodies <= 8 ? 8 : n_bodies <= 16 ? 16 : 32;
    // RSX_LANES_PER_ENV=64 forces the "one wavefront per env" layout (for A/B measurements)
    if (const char* s = std::getenv("RSX_LANES_PER_ENV")) {
        int v = std::atoi(s);
        if ((v == 8 || v == 16 || v == 32 || v == 64) && v >= L) L = v;
    }
    return L;
}

dim3 grid_for(const rsx_sim* h) {
    const int G = 64 / h->L;
    const int tiles = (h->P.num_envs + G - 1) / G;
    return dim3((unsigned)(((tiles + 7) / 8) * 8));
}

#ifdef RSX_TIMING
unsigned long long* g_dbg = nullptr;  // development builds: s_memtime stamps
#endif

// debugging aid (rsx_check_finite / RSX_DEBUG_FINITE=1): counts the non-finite floats of a buffer
__global__ void count_nonfinite_kernel(const float* __restrict__ p, size_t n, unsigned long long* out) {
    unsigned long long bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        bad += !__builtin_isfinite(p[i]);
    if (bad) atomicAdd(out, bad);
}

// adds the per-block-group partial episode counters into metrics[1..7] and clears them (see metric_slot);
// stream-ordered after the step launches whose counts it collects.  One wave; lane = counter.
__global__ void fold_metrics_kernel(unsigned long long* __restrict__ metrics, unsigned long long* __restrict__ slots) {
    const int i = threadIdx.x;
    if (i < 1 || i >= RSX_METRICS) return;   // metrics[0] (env-steps) is kept by the step kernels directly
    unsigned long long sum = 0;
    for (int s = 0; s < MSLOTS; ++s) { sum += slots[(size_t)s * RSX_METRICS + i]; slots[(size_t)s * RSX_METRICS + i] = 0ull; }
    metrics[i] += sum;
}

Buffers buffers_of(const rsx_sim* h, const float* actions) {
    Buffers b;
    b.state = h->d_state; b.aux = h->d_aux; b.obs = h->d_obs; b.final_obs = h->d_final_obs;
    b.flags = h->d_flags; b.cmds = h->d_cmds; b.actions = actions; b.metrics = h->d_metrics; b.mslots = h->d_mslots;
    b.pcache = h->d_pcache; b.pcstats = h->d_pcstats;
#ifdef RSX_TIMING
    b.dbg = g_dbg;
#endif
    return b;
}

// Kernel variants: the common team sizes get the robot count as a template constant (pair
// loops unrolled); anything else runs the generic variant of its lane-group width.
void pick_variant(rsx_sim* h) {
    const int N = h->P.n_robots;
    h->NR = 0;
    if (h->P.kind == RSX_KIND_VSS && N == 6 && (h->L == 8 || h->L == 16) && h->P.n_blue == 3) h->NR = 6;   // 16: RSX_LANES_PER_ENV=16 (four envs per wave)
    if (h->P.kind == RSX_KIND_VSS && N == 10 && h->L == 16 && h->P.n_blue == 5) h->NR = 10;   // 5v5 field
    if (h->P.kind == RSX_KIND_SSL && N == 7 && (h->L == 8 || h->L == 16)) h->NR = 7;
    if (h->P.kind == RSX_KIND_SSL && N == 12 && h->L == 16) h->NR = 12;   // 6v6 (field_type 0, ssl/README.md:4)
    if (h->P.kind == RSX_KIND_SSL && N == 22 && h->L == 32) h->NR = 22;
}

// hot arguments first (preloaded into SGPRs, see RSX_HOT_ARGS), then the by-value structs
#define RSX_LAUNCH_SIM(kernel, P, b) rsx_launch((kernel), grid, dim3(64), 0, s, (b).state, state_out, (b).cmds, (b).flags, \
                                                        (P).num_envs, RSX_HOT_DIM((P).state_dim, (P).row_stride, (P).num_envs), (int)(grid.x >> 3), rand_tick, (P), (b))
#define RSX_LAUNCH(kernel, P, b, n) rsx_launch((kernel), grid, dim3(64), 0, s, (b).state, (b).aux, (b).actions, (b).flags, \
                                                       (P).num_envs, RSX_HOT_DIM((P).state_dim, (P).row_stride, (P).num_envs), (int)(grid.x >> 3), (n), (P), (b))
// the same with `extra` helper workgroups behind the tile workgroups (the tile map still sees the tile grid)
#define RSX_LAUNCH_X(kernel, P, b, n, extra) rsx_launch((kernel), dim3(grid.x + (unsigned)(extra)), dim3(64), 0, s, (b).state, (b).aux, (b).actions, \
                                                                (b).flags, (P).num_envs, RSX_HOT_DIM((P).state_dim, (P).row_stride, (P).num_envs), (int)(grid.x >> 3), (n), (P), (b))

template <int KIND>
void launch_sim_k(const rsx_sim* h, const Params& P_, float* state_out, int rand_tick, hipStream_t s,
                  const float* cmds_src, float* mirror) {
    const dim3 grid = grid_for(h);
    Buffers b = buffers_of(h, nullptr);
    if (cmds_src) b.cmds = cmds_src;                      // commands straight from pinned host memory (small batches)
    b.flags = reinterpret_cast<uint8_t*>(mirror);         // the raw step's fourth pointer slot: second copy of the new state, or null
    if (KIND == RSX_KIND_VSS && h->NR == 6 && h->L == 8) { RSX_LAUNCH_SIM((sim_step_kernel<KIND, 8, (KIND == RSX_KIND_VSS ? 6 : 0)>), P_, b); return; }
    if (KIND == RSX_KIND_VSS && h->NR == 10) { RSX_LAUNCH_SIM((sim_step_kernel<KIND, 16, (KIND == RSX_KIND_VSS ? 10 : 0)>), P_, b); return; }
    if (KIND == RSX_KIND_SSL && h->NR == 7 && h->L == 8) { RSX_LAUNCH_SIM((sim_step_kernel<KIND, 8, (KIND == RSX_KIND_SSL ? 7 : 0)>), P_, b); return; }
    if (KIND == RSX_KIND_SSL && h->NR == 12) { RSX_LAUNCH_SIM((sim_step_kernel<KIND, 16, (KIND == RSX_KIND_SSL ? 12 : 0)>), P_, b); return; }
    if (KIND == RSX_KIND_SSL && h->NR == 22) { RSX_LAUNCH_SIM((sim_step_kernel<KIND, 32, (KIND == RSX_KIND_SSL ? 22 : 0)>), P_, b); return; }
    switch (h->L) {
        case 8: RSX_LAUNCH_SIM((sim_step_kernel<KIND, 8, 0>), P_, b); break;
        case 16: RSX_LAUNCH_SIM((sim_step_kernel<KIND, 16, 0>), P_, b); break;
        case 32: RSX_LAUNCH_SIM((sim_step_kernel<KIND, 32, 0>), P_, b); break;
        default: RSX_LAUNCH_SIM((sim_step_kernel<KIND, 64, 0>), P_, b); break;
    }
}

// state_out: where the new state is written (nullptr = in place)
// rand_tick >= 0: commands drawn in the kernel with Philox key `seed` (rsx_step_dev_random)
void launch_sim(const rsx_sim* h, hipStream_t s, float* state_out = nullptr, int rand_tick = -1, uint64_t seed = 0,
                const float* cmds_src = nullptr, float* mirror = nullptr) {
    if (!state_out) state_out = h->d_state;
    Params P = h->P;
    if (rand_tick >= 0) { P.key0 = (uint32_t)seed; P.key1 = (uint32_t)(seed >> 32); P.env_id_base = 0; }
    if (h->P.kind == RSX_KIND_VSS) launch_sim_k<RSX_KIND_VSS>(h, P, state_out, rand_tick, s, cmds_src, mirror);
    else launch_sim_k<RSX_KIND_SSL>(h, P, state_out, rand_tick, s, cmds_src, mirror);
}

// teleport of rsim.py:52-75 from device arrays: one thread per env, rows are coalesced across threads
__global__ void reset_dev_kernel(float* __restrict__ st, const float* __restrict__ ball, const float* __restrict__ blue,
                                 const float* __restrict__ yellow, const uint8_t* __restrict__ mask, int B, int S_, int rows,
                                 int rs, int nb, int ny, float r_ball) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B || (mask && !mask[e])) return;
    const size_t S = (size_t)S_;   // floats per row
    for (int f = 0; f < rows; ++f) st[(size_t)f * S + e] = 0.0f;
    st[0 * S + e] = ball[4 * (size_t)e + 0]; st[1 * S + e] = ball[4 * (size_t)e + 1]; st[2 * S + e] = r_ball;
    st[3 * S + e] = ball[4 * (size_t)e + 2]; st[4 * S + e] = ball[4 * (size_t)e + 3];
    for (int k = 0; k < nb + ny; ++k) {
        const float* src = k < nb ? blue + ((size_t)e * nb + k) * 3 : yellow + ((size_t)e * ny + (k - nb)) * 3;
        const size_t r = (size_t)(5 + rs * k);
        st[(r + 0) * S + e] = src[0]; st[(r + 1) * S + e] = src[1]; st[(r + 2) * S + e] = src[2];
    }
}

// wire format <-> device layout, for the host-format calls of batches too large for the zero-copy path.  One thread per float64 of the
// wire array (consecutive threads = consecutive addresses of the pinned host buffer: full PCIe packets); the device side of each
// access is a 4-byte piece of an SoA row (absorbed by the L2).
//   commands: wire [B][NC] f64 (rsim.py:92-101 / :129-153)  ->  cmds [NC][S] f32
__global__ void wire_cmds_in_kernel(const double* __restrict__ wire, float* __restrict__ cmds, const unsigned B, const unsigned NC, const unsigned S) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * NC) return;
    const unsigned e = i / NC, j = i - e * NC;
    cmds[(size_t)j * S + e] = (float)wire[i];
}
//   state: state [rows][S] f32  ->  wire [B][rows] f64 (get_state() layout, Entities/Frame.py:20-47 / :55-92, + the two internal rows)
__global__ void wire_state_out_kernel(const float* __restrict__ st, double* __restrict__ wire, const unsigned B, const unsigned rows, const unsigned S) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * rows) return;
    const unsigned e = i / rows, f = i - e * rows;
    wire[i] = (double)st[(size_t)f * S + e];
}

template <int KIND, int TASK, int NRS, int MODE>
void launch_task_m(const rsx_sim* h, const float* actions, int n_steps, hipStream_t s) {
    const Buffers b = buffers_of(h, actions);
    if (TASK == RSX_TASK_VSS_V0 && (MODE == MODE_STEP || MODE == MODE_ROLLOUT) && h->epl && h->NR == 6 && h->L == 8) {
        launch_vss_epl(MODE == MODE_ROLLOUT, h->P, b, n_steps, s);
        return;
    }
    if (TASK == RSX_TASK_SSL_STATIC_DEFENDERS && (MODE == MODE_STEP || MODE == MODE_ROLLOUT) && h->epl && h->NR == 7 && h->L == 8) {
        launch_ssl_epl(TASK, MODE == MODE_ROLLOUT, h->P, b, n_steps, s);
        return;
    }
    const dim3 grid = grid_for(h);
    if (NRS <= 7 && h->NR == NRS && h->L == 8) {
        // single-step launches of a handle with a placement cache: ceil(B / 64) helper workgroups behind the tiles
        const int helpers = (MODE == MODE_STEP && h->d_pcache) ? (h->P.num_envs + 63) / 64 : 0;
        RSX_LAUNCH_X((task_step_kernel<KIND, 8, TASK, (NRS <= 7 ? NRS : 0), MODE>), h->P, b, n_steps, helpers);
        return;
    }
    if (NRS <= 7 && h->NR == NRS && h->L == 16) { RSX_LAUNCH((task_step_kernel<KIND, 16, TASK, (NRS <= 7 ? NRS : 0), MODE>), h->P, b, n_steps); return; }
    if (TASK == RSX_TASK_SSL_SCRIMMAGE && h->NR == 22 && h->L == 32) {   // 11v11: robot count known at compile time
        if (h->quad && MODE == MODE_STEP) { launch_ssl_quad(h->P, b, n_steps, s); return; }
        if (h->big && (MODE == MODE_STEP || MODE == MODE_ROLLOUT)) { launch_scrimmage_big(MODE == MODE_ROLLOUT, h->P, b, n_steps, s); return; }
        RSX_LAUNCH((task_step_kernel<KIND, 32, TASK, (TASK == RSX_TASK_SSL_SCRIMMAGE ? 22 : 0), MODE>), h->P, b, n_steps);
        return;
    }
    if (TASK == RSX_TASK_VSS_V0 && h->NR == 10 && h->L == 16) {   // VSS-v0 on the 5v5 field
        RSX_LAUNCH((task_step_kernel<KIND, 16, TASK, (TASK == RSX_TASK_VSS_V0 ? 10 : 0), MODE>), h->P, b, n_steps);
        return;
    }
    switch (h->L) {
        case 8: RSX_LAUNCH((task_step_kernel<KIND, 8, TASK, 0, MODE>), h->P, b, n_steps); break;
        case 16: RSX_LAUNCH((task_step_kernel<KIND, 16, TASK, 0, MODE>), h->P, b, n_steps); break;
        case 32: RSX_LAUNCH((task_step_kernel<KIND, 32, TASK, 0, MODE>), h->P, b, n_steps); break;
        default: RSX_LAUNCH((task_step_kernel<KIND, 64, TASK, 0, MODE>), h->P, b, n_steps); break;
    }
}

template <int KIND, int TASK, int NRS>
void launch_task_k(const rsx_sim* h, const float* actions, int n_steps, int mode, hipStream_t s) {
    switch (mode) {
        case MODE_STEP: launch_task_m<KIND, TASK, NRS, MODE_STEP>(h, actions, n_steps, s); break;   // n_steps = 1 | flags
        case MODE_ROLLOUT: launch_task_m<KIND, TASK, NRS, MODE_ROLLOUT>(h, nullptr, n_steps, s); break;
        case MODE_RESET: launch_task_m<KIND, TASK, NRS, MODE_RESET>(h, nullptr, 1, s); break;
        default: launch_task_m<KIND, TASK, NRS, MODE_REFRESH>(h, nullptr, 1, s); break;
    }
}

// mode: MODE_STEP (one step, optional fed actions), MODE_ROLLOUT (n_steps in one launch),
// MODE_RESET, MODE_REFRESH
// tasks whose team sizes are fixed by the task: one variant each (8 lanes per env, exact robot count)
template <int TASK, int NRS, int MODE>
void launch_fixed_m(const rsx_sim* h, const float* actions, int n_steps, hipStream_t s) {
    const dim3 grid = grid_for(h);
    const Buffers b = buffers_of(h, actions);
    if ((MODE == MODE_STEP || MODE == MODE_ROLLOUT) && h->epl) { launch_ssl_epl(TASK, MODE == MODE_ROLLOUT, h->P, b, n_steps, s); return; }
    RSX_LAUNCH((task_step_kernel<RSX_KIND_SSL, 8, TASK, NRS, MODE>), h->P, b, n_steps);
}
template <int TASK, int NRS>
void launch_fixed(const rsx_sim* h, const float* actions, int n_steps, int mode, hipStream_t s) {
    switch (mode) {
        case MODE_STEP: launch_fixed_m<TASK, NRS, MODE_STEP>(h, actions, n_steps, s); break;
        case MODE_ROLLOUT: launch_fixed_m<TASK, NRS, MODE_ROLLOUT>(h, nullptr, n_steps, s); break;
        case MODE_RESET: launch_fixed_m<TASK, NRS, MODE_RESET>(h, nullptr, 1, s); break;
        default: launch_fixed_m<TASK, NRS, MODE_REFRESH>(h, nullptr, 1, s); break;
    }
}

void launch_task(const rsx_sim* h, const float* actions, int n_steps, int mode, hipStream_t s) {
    switch (h->P.task) {
        case RSX_TASK_VSS_V0: launch_task_k<RSX_KIND_VSS, RSX_TASK_VSS_V0, 6>(h, actions, n_steps, mode, s); break;
        case RSX_TASK_SSL_STATIC_DEFENDERS: launch_task_k<RSX_KIND_SSL, RSX_TASK_SSL_STATIC_DEFENDERS, 7>(h, actions, n_steps, mode, s); break;
        case RSX_TASK_SSL_DRIBBLING: launch_fixed<RSX_TASK_SSL_DRIBBLING, 5>(h, actions, n_steps, mode, s); break;
        case RSX_TASK_SSL_CONTESTED: launch_fixed<RSX_TASK_SSL_CONTESTED, 2>(h, actions, n_steps, mode, s); break;
        case RSX_TASK_SSL_SCRIMMAGE: case RSX_TASK_SSL_SCRIMMAGE_CROWDED:
            launch_task_k<RSX_KIND_SSL, RSX_TASK_SSL_SCRIMMAGE, 22>(h, actions, n_steps, mode, s); break;
        default: launch_fixed<RSX_TASK_SSL_PASS_ENDURANCE, 2>(h, actions, n_steps, mode, s); break;
    }
}

// workgroups of the handle's stepping launches (MODE_STEP / MODE_ROLLOUT), mirroring the dispatch above: what the
// per-workgroup tick slots of a device-keyed handle are sized and kept in sync by (rsx_kernels.hpp: step_tick)
int step_grid(const rsx_sim* h, int mode) {
    const int B = h->P.num_envs;
    if (h->epl) return epl_grid(B);
    if (h->quad && mode == MODE_STEP) return ssl_quad_grid(B);
    int g = (int)grid_for(h).x;
    if (mode == MODE_STEP && h->d_pcache) g += (B + 63) / 64;   // placement helpers behind the tiles
    return g;
}
uint32_t* tick_words(const rsx_sim* h) { return reinterpret_cast<uint32_t*>(h->d_metrics); }

// slots [from, to) := value, or := slot 0 (copy != 0).  Stream-ordered between two stepping launches.
__global__ void tick_fill_kernel(uint32_t* __restrict__ slots, int from, int to, uint32_t value, int copy) {
    const int i = from + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < to) slots[i] = copy ? __atomic_load_n(&slots[0], __ATOMIC_RELAXED) : value;
}
void tick_fill(const rsx_sim* h, int from, int to, uint32_t value, int copy, hipStream_t s) {
    if (to <= from) return;
    rsx_launch(tick_fill_kernel, dim3((unsigned)((to - from + 255) / 256)), dim3(256), 0, s,
                       tick_words(h) + TICK_SLOT_WORD0, from, to, value, copy);
}

// Makes the handle's device current for the duration of one API call and puts the caller's device
// back on exit: a C-ABI call must not change the thread's current HIP device (which is also
// torch's current device) behind the caller's back.  The per-step calls pay one hipGetDevice.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int enter(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev == device) return RSX_OK;
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return fail(RSX_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
        switched = prev >= 0;
        return RSX_OK;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// (kernel launches report through rsx_launch's own record, rsx_launch.hpp — reset on the way in; the thread's HIP last-error slot, which
// the caller's other HIP work shares, is neither read nor cleared here)
#define RSX_ENTER(h)                                              \
    if (!(h)) return fail(RSX_ERR_ARG, "null handle");            \
    DeviceGuard _guard;                                           \
    if (int _rc = _guard.enter((h)->device)) return _rc;          \
    (void)launch_status()

#define RSX_ENTER_TASK(h)                                                                        \
    RSX_ENTER(h);                                                                                \
    (h)->host_state_valid = false; /* every task call may change the state */                    \
    if ((h)->P.task == RSX_TASK_NONE) return fail(RSX_ERR_STATE, "no task attached (rsx_task_attach)")

// stepping before any reset would run on the dummy line-up with episode id 0xFFFFFFFF
#define RSX_NEED_RESET(h) \
    if (!(h)->task_ready) return fail(RSX_ERR_STATE, "rsx_task_reset / rsx_task_reset_to must come before the first step")

// host f64 AoS [B][S'] <-> device f32 SoA [S'][B]
int upload_state(rsx_sim* h, const std::vector<float>& soa, hipStream_t s) {
    h->host_state_valid = false;
    HIP_TRY(hipMemcpyAsync(h->d_state, soa.data(), soa.size() * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RSX_OK;
}
int download_state(rsx_sim* h, std::vector<float>& soa, hipStream_t s) {
    soa.resize((size_t)(h->P.state_dim + X_ROWS) * h->P.row_stride);
    HIP_TRY(hipMemcpyAsync(soa.data(), h->d_state, soa.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RSX_OK;
}

// write the teleport of rsim.py:52-75 into a host SoA copy
void apply_reset(const rsx_sim* h, std::vector<float>& soa, const double* ball, const double* blue,
                 const double* yellow, const uint8_t* mask) {
    const Params& P = h->P;
    const size_t B = (size_t)P.num_envs, S = (size_t)P.row_stride;   // soa: [rows][S], the device layout
    for (size_t e = 0; e < B; ++e) {
        if (mask && !mask[e]) continue;
        for (int f = 0; f < P.state_dim + X_ROWS; ++f) soa[(size_t)f * S + e] = 0.0f;
        const double* bl = ball + 4 * e;
        soa[0 * S + e] = (float)bl[0]; soa[1 * S + e] = (float)bl[1]; soa[2 * S + e] = (float)h->M.field[6];
        soa[3 * S + e] = (float)bl[2]; soa[4 * S + e] = (float)bl[3];
        for (int k = 0; k < P.n_robots; ++k) {
            const double* src = k < P.n_blue ? blue + ((size_t)e * P.n_blue + k) * 3
                                             : yellow + ((size_t)e * P.n_yellow + (k - P.n_blue)) * 3;
            const size_t r = (size_t)(5 + h->M.rs * k);
            soa[(r + 0) * S + e] = (float)src[0];
            soa[(r + 1) * S + e] = (float)src[1];
            soa[(r + 2) * S + e] = (float)src[2];
        }
    }
}

void free_all(rsx_sim* h) {
    if (h->pin_cmds) (void)hipHostFree(h->pin_cmds);
    if (h->pin_state) (void)hipHostFree(h->pin_state);
    h->pin_cmds = h->pin_state = nullptr;
    if (h->wire_cmds) (void)hipHostFree(h->wire_cmds);
    if (h->wire_state) (void)hipHostFree(h->wire_state);
    h->wire_cmds = h->wire_state = nullptr;
    if (h->alt_alloc) (void)hipFree(h->alt_alloc);   // (not d_state_alt: after an odd number of flips that is a pointer INTO arena_sim)
    h->alt_alloc = h->d_state_alt = nullptr;
    if (h->d_check) (void)hipFree(h->d_check);
    h->d_check = nullptr;
    if (h->arena_sim) (void)hipFree(h->arena_sim);
    if (h->arena_task) (void)hipFree(h->arena_task);
    h->arena_sim = h->arena_task = nullptr;
}

size_t align_up(size_t n) { return (n + 255) & ~(size_t)255; }

// Floats between the rows of the [rows][B] arrays (state, commands, per-env scalars) beyond B.  With rows exactly B floats apart
// and B a power of two — every batch size anybody benchmarks — row f of an env sits at the same address modulo a large power of
// two for every f, and the ~110 read and write streams of a large-batch launch (one per row) walk the same DRAM banks in step.
// A pad of 64 KB + 256 B per row takes them apart: 4 M envs VSS-v0 150.7 -> 139.3 ps per env-step, 1v6 173.3 -> 150.8; 1 M envs 1v6
// 163.5 -> 152.6; nothing at 262 144 envs and below (the arrays sit in the memory-side cache), where the rows stay dense
// (profiles/r05_row_stride.txt: pads from 256 B to 1 MB; exactly 256 KB is the worst, 64 KB + 256 B and 256 KB + 256 B the best).
// A multiple of 64 floats (rows stay 256-byte aligned; it travels in 16 bits of a hot kernel argument: RSX_HOT_DIM).
// RSX_ROW_PAD=<floats> overrides (tests run every kernel family with padded rows at small batches).
constexpr int RSX_ROW_PAD_MIN_ENVS = 786432;   // (524 288 envs measured: no gain yet)
int row_pad_for(int num_envs) {
    // 64 KB + 256 B up to 1.5 M envs, 256 KB + 256 B beyond: at 2 M envs the smaller pad is no better than none (us per step, pads 0 /
    // 16 448 / 65 600 floats: 1 M envs VSS 164 / 148 / 155, 1v6 169 / 157 / 166; 2 M envs VSS 312 / 314 / 286; 4 M envs VSS 638 / 587 / 585,
    // 1v6 730 / 633 / 623)
    long pad = num_envs < RSX_ROW_PAD_MIN_ENVS ? 0 : num_envs < 1572864 ? 16448 : 65600;
    if (const char* v = std::getenv("RSX_ROW_PAD")) pad = std::atol(v);
    if (pad < 0) pad = 0;
    pad = (pad + 63) / 64 * 64;
    if (pad > 65535l * 64) pad = 65535l * 64;
    return (int)pad;
}

}  // namespace

// state rows, and with a task attached observations, rewards and the cumulative info rows
static int check_finite_impl(rsx_sim* h, int64_t* n_bad, hipStream_t s) {
    if (!h->d_check) HIP_TRY(hipMalloc((void**)&h->d_check, sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(h->d_check, 0, sizeof(unsigned long long), s));
    const size_t B = (size_t)h->P.num_envs, S = (size_t)h->P.row_stride;   // (the pad columns hold zeros)
    auto scan = [&](const float* p, size_t n) {
        const unsigned blocks = (unsigned)std::min<size_t>(2048, (n + 255) / 256);
        rsx_launch(count_nonfinite_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, p, n, h->d_check);
    };
    scan(h->d_state, (size_t)(h->P.state_dim + X_ROWS) * S);
    if (h->P.task != RSX_TASK_NONE) {
        scan(h->d_obs, B * (size_t)h->P.obs_dim);
        scan(h->d_aux + (size_t)ROW_REWARD * S, B);
        scan(h->d_aux + (size_t)ROW_INFO * S, S * (size_t)h->M.info_dim);
    }
    HIP_TRY(launch_status());
    unsigned long long bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, h->d_check, sizeof(bad), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *n_bad = (int64_t)bad;
    return RSX_OK;
}

// RSX_DEBUG_FINITE=1: every stepping call is followed by the scan (synchronous: a debugging mode)
static int debug_finite(rsx_sim* h, hipStream_t s, const char* where) {
    static const bool on = std::getenv("RSX_DEBUG_FINITE") != nullptr && std::getenv("RSX_DEBUG_FINITE")[0] == '1';
    if (!on) return RSX_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
    else if (cs != hipStreamCaptureStatusNone) return fail(RSX_ERR_STATE, "RSX_DEBUG_FINITE=1 scans synchronously and cannot run inside a stream capture");
    int64_t bad = 0;
    if (int rc = check_finite_impl(h, &bad, s)) return rc;
    if (bad) return fail(RSX_ERR_STATE, std::string("RSX_DEBUG_FINITE: ") + std::to_string(bad) + " non-finite value(s) after " + where);
    return RSX_OK;
}

extern "C" {

#ifdef RSX_TIMING
int rsx_dbg_set(unsigned long long* p) { g_dbg = p; return 0; }
#endif

int rsx_abi_version(void) { return RSX_ABI_VERSION; }
const char* rsx_last_error(void) { return g_err.c_str(); }

int rsx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rsx_create(rsx_sim** out, int kind, int field_type, int n_blue, int n_yellow, int time_step_ms,
               int num_envs, int device_id) {
    if (!out) return fail(RSX_ERR_ARG, "out is null");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(RSX_ERR_NO_DEVICE, "no HIP device visible: librsx_hip has no CPU path");
    if (device_id < 0 || device_id >= ndev) return fail(RSX_ERR_ARG, "device_id out of range");
    rsx_sim* h = new rsx_sim();
    if (derive_model(kind, field_type, n_blue, n_yellow, time_step_ms, num_envs, h->P, h->M)) {
        delete h;
        return fail(RSX_ERR_ARG, "bad simulator configuration (kind / field_type / robot counts / time step / num_envs)");
    }
    h->P.row_stride = num_envs + row_pad_for(num_envs);
    h->device = device_id;
    h->field_type = field_type; h->time_step_ms = time_step_ms;
    h->L = pick_lanes(h->P.n_robots + 1);
    pick_variant(h);
    auto bail = [&](hipError_t e, const char* what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        free_all(h); delete h;
        return fail(RSX_ERR_HIP, m);
    };
    hipError_t e;
    DeviceGuard guard;
    if (guard.enter(device_id)) { free_all(h); delete h; return RSX_ERR_HIP; }
    const size_t B = (size_t)num_envs, S = (size_t)h->P.row_stride;
    const size_t sbytes = (size_t)(h->P.state_dim + X_ROWS) * S * sizeof(float);
    const size_t cbytes = (size_t)h->P.n_robots * h->M.cmd_dim * S * sizeof(float);
    if (sbytes >= ((size_t)1 << 32)) {   // the kernels address a row of the state with a 32-bit byte offset (rsx_kernels.hpp: at_byte)
        free_all(h); delete h;
        return fail(RSX_ERR_ARG, "num_envs too large: the state array would reach 4 GB (see rsx.h, limits)");
    }
    if ((e = hipMalloc((void**)&h->arena_sim, align_up(sbytes) + align_up(cbytes))) != hipSuccess) return bail(e, "hipMalloc(state+cmds)");
    h->d_state = (float*)h->arena_sim;
    h->d_cmds = (float*)(h->arena_sim + align_up(sbytes));
    if ((e = hipMemset(h->d_cmds, 0, cbytes)) != hipSuccess) return bail(e, "hipMemset(cmds)");
    if ((e = hipHostMalloc((void**)&h->pin_cmds, cbytes ? cbytes : 4, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc(cmds)");
    if ((e = hipHostMalloc((void**)&h->pin_state, sbytes, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc(state)");
    if (num_envs > RSX_ZERO_COPY_MAX_ENVS && !std::getenv("RSX_NO_WIRE_PATH")) {
        // larger batches: pinned buffers in the wire format, converted on the device (see rsx_sim::wire_cmds)
        const size_t wc = B * (size_t)h->P.n_robots * h->M.cmd_dim * sizeof(double), ws = B * (size_t)(h->P.state_dim + X_ROWS) * sizeof(double);
        void *dc = nullptr, *ds = nullptr;
        if (B * (size_t)(h->P.state_dim + X_ROWS) < ((size_t)1 << 32) &&
            hipHostMalloc((void**)&h->wire_cmds, wc, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&h->wire_state, ws, hipHostMallocDefault) == hipSuccess &&
            hipHostGetDevicePointer(&dc, h->wire_cmds, 0) == hipSuccess && hipHostGetDevicePointer(&ds, h->wire_state, 0) == hipSuccess) {
            h->wire_cmds_dev = (double*)dc; h->wire_state_dev = (double*)ds;
        } else {   // no pinned memory to be had, or not addressable by the device: the staging path below still works
            (void)hipGetLastError();
            if (h->wire_cmds) (void)hipHostFree(h->wire_cmds);
            if (h->wire_state) (void)hipHostFree(h->wire_state);
            h->wire_cmds = h->wire_state = nullptr;
        }
    }
    if (num_envs <= RSX_ZERO_COPY_MAX_ENVS && !std::getenv("RSX_NO_ZERO_COPY")) {
        // small batches (the robosim-shaped single-env objects): the raw step reads its commands from, and mirrors the
        // new state into, the pinned host buffers directly — if the device can address them
        void *dc = nullptr, *ds = nullptr;
        if (hipHostGetDevicePointer(&dc, h->pin_cmds, 0) == hipSuccess && hipHostGetDevicePointer(&ds, h->pin_state, 0) == hipSuccess) {
            h->pin_cmds_dev = (float*)dc; h->pin_state_dev = (float*)ds;
        } else {
            (void)hipGetLastError();
        }
    }
    // the adapter's dummy line-up, rsim.py:20-24
    std::vector<float> soa((size_t)(h->P.state_dim + X_ROWS) * S, 0.0f);
    for (size_t i = 0; i < B; ++i) {
        soa[2 * S + i] = (float)h->M.field[6];
        for (int k = 0; k < h->P.n_robots; ++k) {
            const int j = k < n_blue ? k + 1 : k - n_blue + 1;
            soa[(size_t)(5 + h->M.rs * k) * S + i] = (float)((k < n_blue ? -0.2 : 0.2) * j);
        }
    }
    if ((e = hipMemcpy(h->d_state, soa.data(), sbytes, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(state)");
    // the memsets above ran on the null stream; callers step on their own (possibly non-blocking)
    // streams, which do not order themselves behind it
    if ((e = hipDeviceSynchronize()) != hipSuccess) return bail(e, "hipDeviceSynchronize");
    *out = h;
    return RSX_OK;
}

int rsx_destroy(rsx_sim* h) {
    if (!h) return RSX_OK;
    DeviceGuard guard;
    (void)guard.enter(h->device);
    free_all(h);
    delete h;
    return RSX_OK;
}

int rsx_get_field_params(const rsx_sim* h, double out[RSX_FIELD_PARAMS]) {
    if (!h || !out) return fail(RSX_ERR_ARG, "null argument");
    std::memcpy(out, h->M.field, sizeof(double) * RSX_FIELD_PARAMS);
    return RSX_OK;
}

int rsx_reset(rsx_sim* h, const double* ball, const double* blue, const double* yellow,
              const uint8_t* env_mask, void* stream) {
    RSX_ENTER(h);
    if (!ball || (h->P.n_blue && !blue) || (h->P.n_yellow && !yellow)) return fail(RSX_ERR_ARG, "null placement array");
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> soa;
    if (env_mask) { if (int rc = download_state(h, soa, s)) return rc; }
    else soa.assign((size_t)(h->P.state_dim + X_ROWS) * h->P.row_stride, 0.0f);
    apply_reset(h, soa, ball, blue, yellow, env_mask);
    return upload_state(h, soa, s);
}

// the wire-format step of a large batch: commands from h->wire_cmds, new state into h->wire_state (when the host copy is trusted)
static int step_wire_impl(rsx_sim* h, hipStream_t s) {
    const Params& P = h->P;
    const unsigned B = (unsigned)P.num_envs, S = (unsigned)P.row_stride, NC = (unsigned)(P.n_robots * h->M.cmd_dim), rows = (unsigned)(P.state_dim + X_ROWS);
    h->host_state_valid = false;
    rsx_launch(wire_cmds_in_kernel, dim3((B * NC + 255) / 256), dim3(256), 0, s, h->wire_cmds_dev, h->d_cmds, B, NC, S);
    launch_sim(h, s);
    if (h->host_state_cache)
        rsx_launch(wire_state_out_kernel, dim3((B * rows + 255) / 256), dim3(256), 0, s, h->d_state, h->wire_state_dev, B, rows, S);
    HIP_TRY(launch_status());
    HIP_TRY(hipStreamSynchronize(s));
    h->host_state_valid = h->host_state_cache;
    return RSX_OK;
}

int rsx_wire_buffers(rsx_sim* h, double** cmds, double** state) {
    if (!h) return fail(RSX_ERR_ARG, "null handle");
    if (!h->wire_cmds) return fail(RSX_ERR_STATE, "this handle has no wire-format buffers (batches of at most 64 envs step through rsx_step_state without copies)");
    if (cmds) *cmds = h->wire_cmds;
    if (state) *state = h->wire_state;
    return RSX_OK;
}

int rsx_step_wire(rsx_sim* h, void* stream) {
    RSX_ENTER(h);
    if (!h->wire_cmds) return fail(RSX_ERR_STATE, "this handle has no wire-format buffers (rsx_wire_buffers)");
    if (!h->host_state_cache) {   // a device view was handed out: nothing mirrors the state unasked any more, but this call promises it
        h->host_state_cache = true;
        const int rc = step_wire_impl(h, (hipStream_t)stream);
        h->host_state_cache = false; h->host_state_valid = false;
        return rc;
    }
    return step_wire_impl(h, (hipStream_t)stream);
}

int rsx_step(rsx_sim* h, const double* cmds, void* stream) {
    RSX_ENTER(h);
    if (!cmds) return fail(RSX_ERR_ARG, "cmds is null");
    hipStream_t s = (hipStream_t)stream;
    const Params& P = h->P;
    const size_t B = (size_t)P.num_envs, S = (size_t)P.row_stride, NC = (size_t)P.n_robots * h->M.cmd_dim;
    if (h->wire_cmds) {   // large batch: one contiguous copy into the pinned wire buffer, the conversion runs on the device
        if (cmds != h->wire_cmds) std::memcpy(h->wire_cmds, cmds, B * NC * sizeof(double));
        return step_wire_impl(h, s);
    }
    for (size_t e = 0; e < B; ++e)
        for (size_t j = 0; j < NC; ++j) h->pin_cmds[j * S + e] = (float)cmds[e * NC + j];
    h->host_state_valid = false;
    if (h->pin_cmds_dev) {   // small batch: no copies, the kernel talks to the pinned buffers
        launch_sim(h, s, nullptr, -1, 0, h->pin_cmds_dev, h->host_state_cache ? h->pin_state_dev : nullptr);
        HIP_TRY(launch_status());
    } else {
        HIP_TRY(hipMemcpyAsync(h->d_cmds, h->pin_cmds, NC * S * sizeof(float), hipMemcpyHostToDevice, s));
        launch_sim(h, s);
        HIP_TRY(launch_status());
        if (h->host_state_cache)
            HIP_TRY(hipMemcpyAsync(h->pin_state, h->d_state, (size_t)(P.state_dim + X_ROWS) * S * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    h->host_state_valid = h->host_state_cache;
    return RSX_OK;
}

static int get_state_impl(rsx_sim* h, double* out, int rows, hipStream_t s) {
    const size_t B = (size_t)h->P.num_envs, S = (size_t)h->P.row_stride;
    if (h->wire_state) {   // large batch: the wire-format copy is made on the device; what is left is a copy out of pinned memory
        const int all = h->P.state_dim + X_ROWS;
        if (!h->host_state_valid) {
            rsx_launch(wire_state_out_kernel, dim3(((unsigned)B * all + 255) / 256), dim3(256), 0, s, h->d_state, h->wire_state_dev, (unsigned)B, (unsigned)all, (unsigned)S);
            HIP_TRY(launch_status());
            HIP_TRY(hipStreamSynchronize(s));
            h->host_state_valid = h->host_state_cache;
        }
        if (out == h->wire_state) return RSX_OK;
        if (rows == all) std::memcpy(out, h->wire_state, B * (size_t)all * sizeof(double));
        else for (size_t e = 0; e < B; ++e) std::memcpy(out + e * rows, h->wire_state + e * all, (size_t)rows * sizeof(double));
        return RSX_OK;
    }
    if (!h->host_state_valid) {
        HIP_TRY(hipMemcpyAsync(h->pin_state, h->d_state, (size_t)(h->P.state_dim + X_ROWS) * S * sizeof(float), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        h->host_state_valid = h->host_state_cache;
    }
    const float* soa = h->pin_state;
    for (size_t e = 0; e < B; ++e)
        for (int f = 0; f < rows; ++f) out[e * rows + f] = (double)soa[(size_t)f * S + e];
    return RSX_OK;
}

int rsx_get_state(rsx_sim* h, double* out, void* stream) {
    RSX_ENTER(h);
    if (!out) return fail(RSX_ERR_ARG, "out is null");
    return get_state_impl(h, out, h->P.state_dim, (hipStream_t)stream);
}

int rsx_step_state(rsx_sim* h, const double* cmds, double* state_out, void* stream) {
    if (!state_out) return fail(RSX_ERR_ARG, "state_out is null");
    if (int rc = rsx_step(h, cmds, stream)) return rc;
    return rsx_get_state(h, state_out, stream);
}

int rsx_get_state_full(rsx_sim* h, double* out, void* stream) {
    RSX_ENTER(h);
    if (!out) return fail(RSX_ERR_ARG, "out is null");
    return get_state_impl(h, out, h->P.state_dim + X_ROWS, (hipStream_t)stream);
}

int rsx_set_state(rsx_sim* h, const double* state, void* stream) {
    RSX_ENTER(h);
    if (!state) return fail(RSX_ERR_ARG, "state is null");
    const size_t B = (size_t)h->P.num_envs, S = (size_t)h->P.row_stride;
    const int rows = h->P.state_dim + X_ROWS;
    std::vector<float> soa((size_t)rows * S);   // (zeros in the pad columns)
    for (size_t e = 0; e < B; ++e)
        for (int f = 0; f < rows; ++f) soa[(size_t)f * S + e] = (float)state[e * rows + f];
    return upload_state(h, soa, (hipStream_t)stream);
}

int rsx_dev_view_get(rsx_sim* h, rsx_dev_view* out) {
    if (!h || !out) return fail(RSX_ERR_ARG, "null argument");
    out->num_envs = h->P.num_envs; out->n_robots = h->P.n_robots;
    out->state_dim = h->P.state_dim; out->cmd_dim = h->M.cmd_dim;
    out->state = h->d_state; out->cmds = h->d_cmds; out->row_stride = h->P.row_stride;
    h->host_state_cache = false; h->host_state_valid = false;   // the caller can now write the state behind our back
    return RSX_OK;
}

int rsx_step_dev(rsx_sim* h, void* stream) {
    RSX_ENTER(h);
    h->host_state_valid = false;
    launch_sim(h, (hipStream_t)stream);
    HIP_TRY(launch_status());
    return debug_finite(h, (hipStream_t)stream, "rsx_step_dev");
}

static int ensure_alt(rsx_sim* h) {
    if (h->d_state_alt) return RSX_OK;
    const size_t sbytes = (size_t)(h->P.state_dim + X_ROWS) * h->P.row_stride * sizeof(float);
    HIP_TRY(hipMalloc((void**)&h->alt_alloc, sbytes));
    h->d_state_alt = h->alt_alloc;
    HIP_TRY(hipMemset(h->d_state_alt, 0, sbytes));
    HIP_TRY(hipDeviceSynchronize());
    return RSX_OK;
}

int rsx_state_buffers(rsx_sim* h, float** current, float** other) {
    RSX_ENTER(h);
    if (!current || !other) return fail(RSX_ERR_ARG, "null argument");
    if (int rc = ensure_alt(h)) return rc;
    *current = h->d_state; *other = h->d_state_alt;
    h->host_state_cache = false; h->host_state_valid = false;
    return RSX_OK;
}

int rsx_step_dev_random(rsx_sim* h, int n, uint64_t seed, uint32_t first_tick, void* stream) {
    RSX_ENTER(h);
    if (n < 1) return fail(RSX_ERR_ARG, "n must be >= 1");
    if ((uint64_t)first_tick + (uint64_t)n > 0x7FFFFFFFull) return fail(RSX_ERR_ARG, "tick range exceeds 2^31");
    h->host_state_valid = false;
    for (int i = 0; i < n; ++i) launch_sim(h, (hipStream_t)stream, nullptr, (int)(first_tick + (uint32_t)i), seed);
    HIP_TRY(launch_status());
    return debug_finite(h, (hipStream_t)stream, "rsx_step_dev_random");
}

int rsx_step_dev_flip(rsx_sim* h, void* stream) {
    RSX_ENTER(h);
    {   // which buffer is current is host state: a replayed graph would keep writing the same one
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) (void)hipGetLastError();
        else if (cs != hipStreamCaptureStatusNone) return fail(RSX_ERR_STATE, "rsx_step_dev_flip cannot be captured (the buffer roles are host state); capture rsx_step_dev instead");
    }
    if (int rc = ensure_alt(h)) return rc;
    h->host_state_valid = false;
    launch_sim(h, (hipStream_t)stream, h->d_state_alt);
    HIP_TRY(launch_status());
    std::swap(h->d_state, h->d_state_alt);
    return debug_finite(h, (hipStream_t)stream, "rsx_step_dev_flip");
}

int rsx_reset_dev(rsx_sim* h, const float* ball_dev, const float* blue_dev, const float* yellow_dev,
                  const uint8_t* env_mask_dev, void* stream) {
    RSX_ENTER(h);
    if (!ball_dev || (h->P.n_blue && !blue_dev) || (h->P.n_yellow && !yellow_dev)) return fail(RSX_ERR_ARG, "null placement array");
    h->host_state_valid = false;
    const int B = h->P.num_envs;
    rsx_launch(reset_dev_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->d_state, ball_dev, blue_dev,
                       yellow_dev, env_mask_dev, B, h->P.row_stride, h->P.state_dim + X_ROWS, h->M.rs, h->P.n_blue, h->P.n_yellow, (float)h->M.field[6]);
    HIP_TRY(launch_status());
    return RSX_OK;
}

int rsx_task_attach(rsx_sim* h, int task, uint64_t seed, uint64_t env_id_base, int max_episode_steps) {
    RSX_ENTER(h);
    if (h->P.task != RSX_TASK_NONE) return fail(RSX_ERR_STATE, "a task is already attached");
    // global env ids are 32-bit words of the Philox counter: the whole range of this handle has to fit
    if (env_id_base > 0xFFFFFFFFull || env_id_base + (uint64_t)h->P.num_envs > 0x100000000ull)
        return fail(RSX_ERR_ARG, "env_id_base + num_envs exceeds 2^32 (global env ids are 32-bit)");
    Params P = h->P;
    if (derive_task(task, seed, env_id_base, max_episode_steps, h->M, P))
        return fail(RSX_ERR_ARG, "task does not match the simulator (VSS_V0: VSS, n_blue >= 1; STATIC_DEFENDERS: SSL 1vN; DRIBBLING: SSL 1v4; CONTESTED: SSL 1v1; PASS_ENDURANCE: SSL 2v0; SCRIMMAGE: SSL)");
    if (P.obs_dim > 64) return fail(RSX_ERR_ARG, "observation wider than 64 floats is not supported");
    if (task >= RSX_TASK_SSL_DRIBBLING && task <= RSX_TASK_SSL_PASS_ENDURANCE && h->L != 8)
        return fail(RSX_ERR_ARG, "this task runs with 8 lanes per env only (unset RSX_LANES_PER_ENV)");
    const size_t B = (size_t)P.num_envs, S = (size_t)P.row_stride;
    const size_t n_aux = align_up((size_t)aux_rows(P.n_robots) * S * sizeof(float));
    if (n_aux >= ((size_t)1 << 32) || B * (size_t)P.obs_dim * sizeof(float) >= ((size_t)1 << 32))
        return fail(RSX_ERR_ARG, "num_envs too large for a fused task: the per-env scalar arena or the observation array would reach 4 GB (see rsx.h, limits)");
    const size_t n_obs = align_up(B * P.obs_dim * sizeof(float));
    const size_t n_flags = align_up(3 * B);   // terminated | truncated | the env mask of rsx_task_reset_to (read by the MODE_REFRESH launch only)
    const size_t n_act = align_up(B * h->M.act_dim * sizeof(float));
    // metrics[8] | error word | (256 bytes in) one step-counter slot per workgroup of the largest stepping launch any layout
    // of this batch could use (rsx_kernels.hpp: step_tick; only the first tick_slots are kept in sync)
    const size_t n_met = align_up((size_t)TICK_SLOT_WORD0 * 4 + ((size_t)grid_for(h).x + (B + 63) / 64) * sizeof(uint32_t));
    const size_t n_slots = align_up((size_t)MSLOTS * RSX_METRICS * sizeof(unsigned long long));
    // placement cache: static defenders 1v6 (short episodes: several resetting waves per launch) at latency-bound batches
    const bool pc = !std::getenv("RSX_NO_PCACHE") && h->L == 8 && P.num_envs <= RSX_PCACHE_MAX_ENVS && P.n_sub > 0 &&
                    task == RSX_TASK_SSL_STATIC_DEFENDERS && h->NR == 7;
    const size_t n_pc = pc ? align_up((size_t)2 * (3 * (P.n_robots + 1) + 1) * B * sizeof(float)) : 0;
    const size_t n_pcs = pc && std::getenv("RSX_PCACHE_STATS") ? align_up(2 * sizeof(unsigned long long)) : 0;
    const size_t total = n_aux + 2 * n_obs + n_flags + n_act + n_met + n_slots + n_pc + n_pcs;
    HIP_TRY(hipMalloc((void**)&h->arena_task, total));
    HIP_TRY(hipMemset(h->arena_task, 0, total));
    h->arena_task_bytes = total; h->pcache_bytes = n_pc;
    char* p = h->arena_task;
    h->d_aux = (float*)p; p += n_aux;
    h->d_obs = (float*)p; p += n_obs;
    h->d_final_obs = (float*)p; p += n_obs;
    h->d_flags = (uint8_t*)p; p += n_flags;
    h->d_actions = (float*)p; p += n_act;
    h->d_metrics = (unsigned long long*)p; p += n_met;
    h->d_mslots = (unsigned long long*)p; p += n_slots;
    if (n_pc) {
        h->d_pcache = (float*)p; p += n_pc;
        HIP_TRY(hipMemset(h->d_pcache, 0xFF, n_pc));   // tags 0xFFFFFFFF: no entry is valid yet
    }
    if (n_pcs) h->d_pcstats = (unsigned long long*)p;
    // episode ids start at 0xFFFFFFFF so that the first reset() opens episode 0
    HIP_TRY(hipMemset(h->d_aux + (size_t)ROW_EPISODE * S, 0xFF, B * sizeof(uint32_t)));
    h->P = P;
    // The five registered tasks: which tile layout steps the envs.  Both give identical results; the one-lane-
    // per-env kernel needs enough envs to fill the chip with its long waves.
    h->epl = false;
    h->big = false; h->quad = false;
    if ((task == RSX_TASK_SSL_SCRIMMAGE || task == RSX_TASK_SSL_SCRIMMAGE_CROWDED) && h->NR == 22 && h->L == 32) {
        h->big = P.num_envs >= RSX_BIG_MIN_ENVS;
        // four lanes per env: 32-bit row offsets (arrays below 2 GB), a real time step (the infrared row is rewritten)
        const char* lay = std::getenv("RSX_LAYOUT");
        const size_t rows = (size_t)std::max(P.state_dim + X_ROWS, aux_rows(P.n_robots));
        const bool fits = rows * (size_t)P.row_stride * sizeof(float) < ((size_t)1 << 31) && P.n_sub > 0 && P.n_blue == 11;
        // measured crossovers (profiles/LABBOOK.md): the spread line-up from 32 768 envs, the crowded one (contacts in every
        // sub-step: the six robots of a lane are walked one after the other) from 65 536 (RSX_QUAD_MIN_ENVS_CROWDED);
        // multi-step calls on a crowded handle stay with the 32-lane kernel below 262 144 envs (rsx_task_rollout)
        const int quad_min = task == RSX_TASK_SSL_SCRIMMAGE ? RSX_QUAD_MIN_ENVS : RSX_QUAD_MIN_ENVS_CROWDED;
        h->quad = fits && (lay ? std::strcmp(lay, "quad") == 0 : (quad_min > 0 && P.num_envs >= quad_min));
    }
    const bool fixed_ssl = task == RSX_TASK_SSL_DRIBBLING || task == RSX_TASK_SSL_CONTESTED || task == RSX_TASK_SSL_PASS_ENDURANCE;   // team sizes checked above
    if ((task == RSX_TASK_VSS_V0 && h->NR == 6 && h->L == 8) || (task == RSX_TASK_SSL_STATIC_DEFENDERS && h->NR == 7 && h->L == 8) || fixed_ssl) {
        const char* lay = std::getenv("RSX_LAYOUT");
        if (lay && std::strcmp(lay, "epl") == 0) h->epl = true;
        else if (lay && std::strcmp(lay, "lanes") == 0) h->epl = false;
        else h->epl = P.num_envs >= (task == RSX_TASK_VSS_V0 ? RSX_EPL_MIN_ENVS : task == RSX_TASK_SSL_STATIC_DEFENDERS ? RSX_EPL_MIN_ENVS_SSL
                                     : task == RSX_TASK_SSL_DRIBBLING ? 49152 : 32768);
        // those kernels address rows with 32-bit byte offsets (buffer instructions): arrays of 2 GB and more stay with the lane-group kernels
        const size_t rows = (size_t)std::max(P.state_dim + X_ROWS, aux_rows(P.n_robots));
        if (rows * (size_t)P.row_stride * sizeof(float) >= ((size_t)1 << 31) || (size_t)P.num_envs * P.obs_dim * sizeof(float) >= ((size_t)1 << 31)) h->epl = false;
    }
    h->task_ready = false;
    h->tick_dev = false;
    h->tick_slots = step_grid(h, MODE_STEP);
    h->tick_slots_alloc = std::max(h->tick_slots, (int)grid_for(h).x + (P.num_envs + 63) / 64);
    HIP_TRY(hipDeviceSynchronize());   // null-stream memsets done before any caller stream steps
    return RSX_OK;
}

int rsx_drop_pending_hip_error(void) { return (int)hipGetLastError(); }

int rsx_task_reseed(rsx_sim* h, uint64_t seed, void* stream) {
    RSX_ENTER_TASK(h);
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
    else if (cs != hipStreamCaptureStatusNone) return fail(RSX_ERR_STATE, "rsx_task_reseed changes host state (seed, step counter) and cannot be captured");
    // what rsx_task_attach leaves behind, with the new key: every per-env buffer and counter cleared, episode ids at 0xFFFFFFFF,
    // placement cache empty, step counter 0 (device-keyed handles: every slot), no episode open
    HIP_TRY(hipMemsetAsync(h->arena_task, 0, h->arena_task_bytes, s));
    if (h->d_pcache) HIP_TRY(hipMemsetAsync(h->d_pcache, 0xFF, h->pcache_bytes, s));
    HIP_TRY(hipMemsetAsync(h->d_aux + (size_t)ROW_EPISODE * h->P.row_stride, 0xFF, (size_t)h->P.num_envs * sizeof(uint32_t), s));
    h->P.key0 = (uint32_t)seed; h->P.key1 = (uint32_t)(seed >> 32);
    h->tick = 0; h->P.tick_base = 0;
    h->task_ready = false;
    return RSX_OK;
}

int rsx_task_enable_capture(rsx_sim* h, void* stream) {
    RSX_ENTER_TASK(h);
    // The one place where the thread's pending HIP error is dropped: the usual way to get here is a capture attempt that this library
    // refused (host-keyed handle) and that the caller's framework then aborted — which leaves `invalid argument` in the slot for the
    // NEXT capture to trip over (torch.cuda.graph does).  A set-up call, not a stepping path; rsx.h says so.
    (void)hipGetLastError();
    if (h->tick_dev) return RSX_OK;
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
    else if (cs != hipStreamCaptureStatusNone)
        return fail(RSX_ERR_STATE, "rsx_task_enable_capture must be called BEFORE the capture begins (it writes the step counter once; a captured write would reset it on every replay)");
    tick_fill(h, 0, h->tick_slots_alloc, h->tick, 0, s);
    HIP_TRY(launch_status());
    h->tick_dev = true;
    h->host_state_cache = false; h->host_state_valid = false;   // a replayed graph changes the state without passing through this API
    return RSX_OK;
}

int rsx_task_tick(rsx_sim* h, uint32_t* out, void* stream) {
    RSX_ENTER(h);
    if (!out) return fail(RSX_ERR_ARG, "out is null");
    if (h->P.task == RSX_TASK_NONE) return fail(RSX_ERR_STATE, "no task attached (rsx_task_attach)");
    if (!h->tick_dev) { *out = h->tick; return RSX_OK; }
    uint32_t w[2] = {0, 0};   // slot 0, error word
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(&w[0], tick_words(h) + TICK_SLOT_WORD0, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&w[1], tick_words(h) + TICK_ERR_WORD, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *out = w[0];
    h->tick = w[0];
    if (w[1]) return fail(RSX_ERR_STATE, "step counter exhausted: a launch that would have wrapped it was refused on the device (a handle takes at most 2^32 - 1 fused steps)");
    return RSX_OK;
}

int rsx_task_placement_cache_stats(rsx_sim* h, int64_t out[2], void* stream) {
    RSX_ENTER_TASK(h);
    if (!out) return fail(RSX_ERR_ARG, "out is null");
    out[0] = out[1] = -1;   // -1: no cache on this handle, or the counters are off (RSX_PCACHE_STATS=1 before rsx_task_attach)
    if (!h->d_pcstats) return RSX_OK;
    HIP_TRY(hipMemcpyAsync(out, h->d_pcstats, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return RSX_OK;
}

int rsx_task_layout(rsx_sim* h, char* out, size_t n) {
    if (!h || !out || n == 0) return fail(RSX_ERR_ARG, "null argument");
    if (h->P.task == RSX_TASK_NONE) return fail(RSX_ERR_STATE, "no task attached (rsx_task_attach)");
    const char* name = h->epl ? "one-lane-per-env" : h->quad ? "four-lanes-per-env" : h->big ? "32-lanes-per-env-large-batch"
                     : h->L == 8 ? "8-lanes-per-env" : h->L == 16 ? "16-lanes-per-env" : h->L == 32 ? "32-lanes-per-env" : "64-lanes-per-env";
    std::snprintf(out, n, "%s", name);
    return RSX_OK;
}

int rsx_task_view_get(rsx_sim* h, rsx_task_view* out) {
    if (!h || !out) return fail(RSX_ERR_ARG, "null argument");
    if (h->P.task == RSX_TASK_NONE) return fail(RSX_ERR_STATE, "no task attached (rsx_task_attach)");
    const size_t B = (size_t)h->P.num_envs;
    out->task = h->P.task; out->obs_dim = h->P.obs_dim; out->act_dim = h->M.act_dim;
    out->info_dim = h->M.info_dim; out->max_episode_steps = h->P.max_steps;
    const size_t S = (size_t)h->P.row_stride;
    out->obs = h->d_obs; out->reward = h->d_aux + (size_t)ROW_REWARD * S;
    out->terminated = h->d_flags; out->truncated = h->d_flags + B;
    out->info = h->d_aux + (size_t)ROW_INFO * S; out->final_obs = h->d_final_obs;
    out->steps = (int32_t*)(h->d_aux + (size_t)ROW_STEPS * S); out->actions = h->d_actions;
    out->metrics = (int64_t*)h->d_metrics; out->row_stride = h->P.row_stride;
    return RSX_OK;
}

int rsx_task_reset(rsx_sim* h, void* stream) {
    RSX_ENTER_TASK(h);
    launch_task(h, nullptr, 1, MODE_RESET, (hipStream_t)stream);
    HIP_TRY(launch_status());
    h->task_ready = true;
    return RSX_OK;
}

int rsx_task_reset_to(rsx_sim* h, const double* ball, const double* blue, const double* yellow,
                      const uint8_t* env_mask, void* stream) {
    RSX_ENTER_TASK(h);
    if (int rc = rsx_reset(h, ball, blue, yellow, env_mask, stream)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t B = (size_t)h->P.num_envs;
    // the kernel takes the env mask through the third row of the flags array: `terminated` / `truncated` of the envs the mask leaves
    // alone stay what their last step made them (until round 6 the mask travelled through the `truncated` row, which was cleared
    // afterwards — for EVERY env: found by tests/test_gpu_api_fuzz.py)
    if (env_mask) HIP_TRY(hipMemcpyAsync(h->d_flags + 2 * B, env_mask, B, hipMemcpyHostToDevice, s));
    else HIP_TRY(hipMemsetAsync(h->d_flags + 2 * B, 1, B, s));
    launch_task(h, nullptr, 1, MODE_REFRESH, s);
    HIP_TRY(launch_status());
    HIP_TRY(hipStreamSynchronize(s));   // the host mask / placement arrays may be reused by the caller
    h->task_ready = true;
    return RSX_OK;
}

// The handle's step counter keys the per-step random draws and is one 32-bit word of the Philox counter: a handle that
// has taken 2^32 - 1 fused steps refuses further ones instead of silently replaying its random streams.
// Host-keyed handles (the default) check that here and bake the count into the launch — which is why they refuse to be
// captured: a replayed graph would step with one tick for ever.  Device-keyed handles (rsx_task_enable_capture) pass
// RSX_TICK_DEV instead: the kernels read, check and advance the counter themselves (rsx_kernels.hpp: step_tick).
static int step_prologue(rsx_sim* h, hipStream_t s, uint64_t n, int* flags) {
    *flags = 0;
    if (h->tick_dev) { *flags = RSX_TICK_DEV; return RSX_OK; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();   // (e.g. the legacy stream while another one captures: the launch reports it)
    else if (cs != hipStreamCaptureStatusNone)
        return fail(RSX_ERR_STATE, "this stream is being captured, and the handle's step counter (the key of its per-step random draws) is still a host-side "
                                   "launch argument: a replayed graph would repeat one random stream. Call rsx_task_enable_capture(h, stream) once, before the capture begins");
    if ((uint64_t)h->tick + n > 0xFFFFFFFFull)
        return fail(RSX_ERR_STATE, "step counter exhausted: a handle takes at most 2^32 - 1 fused steps (it keys the per-step random draws); attach a fresh handle with another seed");
    return RSX_OK;
}
// device-keyed handles: slots the launch did not cover (a grid without the placement helpers) follow slot 0
static void tick_resync(const rsx_sim* h, int mode, hipStream_t s) {
    if (h->tick_dev) tick_fill(h, step_grid(h, mode), h->tick_slots, 0u, 1, s);
}

int rsx_task_step(rsx_sim* h, const float* actions_dev, void* stream) {
    RSX_ENTER_TASK(h);
    RSX_NEED_RESET(h);
    int fl = 0;
    if (int rc = step_prologue(h, (hipStream_t)stream, 1, &fl)) return rc;
    h->P.tick_base = h->tick++;
    launch_task(h, actions_dev, 1 | fl, MODE_STEP, (hipStream_t)stream);
    HIP_TRY(launch_status());
    return debug_finite(h, (hipStream_t)stream, "rsx_task_step");
}

int rsx_task_step_n(rsx_sim* h, int n, void* stream) {
    RSX_ENTER_TASK(h);
    RSX_NEED_RESET(h);
    if (n < 1) return fail(RSX_ERR_ARG, "n must be >= 1");
    int fl = 0;
    if (int rc = step_prologue(h, (hipStream_t)stream, (uint64_t)n, &fl)) return rc;
    for (int i = 0; i < n; ++i) { h->P.tick_base = h->tick++; launch_task(h, nullptr, 1 | fl, MODE_STEP, (hipStream_t)stream); }
    HIP_TRY(launch_status());
    return debug_finite(h, (hipStream_t)stream, "rsx_task_step_n");
}

int rsx_task_rollout(rsx_sim* h, int n, void* stream) {
    RSX_ENTER_TASK(h);
    RSX_NEED_RESET(h);
    if (n < 0 || n > RSX_N_STEPS_MASK) return fail(RSX_ERR_ARG, "n must be in 0 .. 2^30 - 1");  // 0 = load + store only (profiling)
    int fl = 0;
    if (int rc = step_prologue(h, (hipStream_t)stream, (uint64_t)n, &fl)) return rc;
    // (a device-keyed four-lane handle always takes the first form: all of its stepping launches then share one grid)
    if (h->quad && n >= 1 && (h->tick_dev || h->P.num_envs >= (h->P.task == RSX_TASK_SSL_SCRIMMAGE ? RSX_QUAD_ROLLOUT_MIN_ENVS : RSX_QUAD_ROLLOUT_MIN_ENVS_CROWDED))) {
        // 11v11 at large batches: n launches of the four-lanes-per-env kernel beat one launch of the 32-lane kernel
        // (262 144 envs, us per step: spread 182 vs 275, crowded 298 vs 340; crowded 131 072: 161 vs 164); same steps, same results
        for (int i = 0; i < n; ++i) { h->P.tick_base = h->tick++; launch_task(h, nullptr, 1 | fl, MODE_STEP, (hipStream_t)stream); }
        HIP_TRY(launch_status());
        return debug_finite(h, (hipStream_t)stream, "rsx_task_rollout");
    }
    h->P.tick_base = h->tick; h->tick += (uint32_t)n;
    launch_task(h, nullptr, n | fl, MODE_ROLLOUT, (hipStream_t)stream);
    tick_resync(h, MODE_ROLLOUT, (hipStream_t)stream);
    HIP_TRY(launch_status());
    return debug_finite(h, (hipStream_t)stream, "rsx_task_rollout");
}

int rsx_check_finite(rsx_sim* h, int64_t* n_bad, void* stream) {
    RSX_ENTER(h);
    if (!n_bad) return fail(RSX_ERR_ARG, "n_bad is null");
    return check_finite_impl(h, n_bad, (hipStream_t)stream);
}

}  // extern "C" (reopened below)

// ---- task checkpoint: everything a fused run needs to continue bit-identically ----
namespace {
struct CkptHeader {
    uint64_t magic;            // "RSXCKPT2"
    int32_t abi, kind, field_rows, task, n_blue, n_yellow, num_envs, state_rows, aux_rows, obs_dim;
    int32_t field_type, time_step_ms, max_steps, model;   // model: RSX_PHYSICS_MODEL of the saving library
    uint32_t key0, key1, env_id_base, tick;
    uint64_t state_bytes, aux_bytes, obs_bytes, flag_bytes;
    int64_t metrics[RSX_METRICS];
};
constexpr uint64_t CKPT_MAGIC = 0x3254504B43585352ull;   // "RSXCKPT2", little endian
CkptHeader ckpt_header(const rsx_sim* h) {
    CkptHeader k{};
    const size_t B = (size_t)h->P.num_envs;
    k.magic = CKPT_MAGIC; k.abi = RSX_ABI_VERSION; k.kind = h->P.kind; k.field_rows = h->M.rs; k.task = h->P.task;
    k.n_blue = h->P.n_blue; k.n_yellow = h->P.n_yellow; k.num_envs = h->P.num_envs;
    k.state_rows = h->P.state_dim + X_ROWS; k.aux_rows = aux_rows(h->P.n_robots); k.obs_dim = h->P.obs_dim;
    k.field_type = h->field_type; k.time_step_ms = h->time_step_ms; k.max_steps = h->P.max_steps; k.model = RSX_PHYSICS_MODEL;
    k.key0 = h->P.key0; k.key1 = h->P.key1; k.env_id_base = h->P.env_id_base; k.tick = h->tick;
    k.state_bytes = (uint64_t)k.state_rows * B * sizeof(float);
    k.aux_bytes = (uint64_t)k.aux_rows * B * sizeof(float);
    k.obs_bytes = (uint64_t)B * k.obs_dim * sizeof(float);
    k.flag_bytes = 2 * B;
    return k;
}
size_t ckpt_size(const CkptHeader& k) { return sizeof(CkptHeader) + k.state_bytes + k.aux_bytes + 2 * k.obs_bytes + k.flag_bytes; }
}  // namespace

extern "C" {

int rsx_task_checkpoint_size(rsx_sim* h, size_t* bytes) {
    if (!h || !bytes) return fail(RSX_ERR_ARG, "null argument");
    if (h->P.task == RSX_TASK_NONE) return fail(RSX_ERR_STATE, "no task attached (rsx_task_attach)");
    *bytes = ckpt_size(ckpt_header(h));
    return RSX_OK;
}

int rsx_task_checkpoint_save(rsx_sim* h, void* blob, size_t bytes, void* stream) {
    RSX_ENTER_TASK(h);
    if (!blob) return fail(RSX_ERR_ARG, "blob is null");
    CkptHeader k = ckpt_header(h);
    if (bytes < ckpt_size(k)) return fail(RSX_ERR_ARG, "blob is smaller than rsx_task_checkpoint_size");
    hipStream_t s = (hipStream_t)stream;
    rsx_launch(fold_metrics_kernel, dim3(1), dim3(64), 0, s, h->d_metrics, h->d_mslots);
    HIP_TRY(launch_status());
    char* p = (char*)blob + sizeof(CkptHeader);
    // (the blob's rows are dense — B floats — whatever the row pad of this handle: it restores into any layout)
    const size_t rowb = (size_t)h->P.num_envs * sizeof(float), pitch = (size_t)h->P.row_stride * sizeof(float);
    HIP_TRY(hipMemcpy2DAsync(p, rowb, h->d_state, pitch, rowb, (size_t)k.state_rows, hipMemcpyDeviceToHost, s)); p += k.state_bytes;
    HIP_TRY(hipMemcpy2DAsync(p, rowb, h->d_aux, pitch, rowb, (size_t)k.aux_rows, hipMemcpyDeviceToHost, s)); p += k.aux_bytes;
    HIP_TRY(hipMemcpyAsync(p, h->d_obs, k.obs_bytes, hipMemcpyDeviceToHost, s)); p += k.obs_bytes;
    HIP_TRY(hipMemcpyAsync(p, h->d_final_obs, k.obs_bytes, hipMemcpyDeviceToHost, s)); p += k.obs_bytes;
    HIP_TRY(hipMemcpyAsync(p, h->d_flags, k.flag_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(k.metrics, h->d_metrics, sizeof(k.metrics), hipMemcpyDeviceToHost, s));
    if (h->tick_dev)   // device-keyed handle: the step counter is slot 0 of the per-workgroup slots (all equal between launches)
        HIP_TRY(hipMemcpyAsync(&k.tick, tick_words(h) + TICK_SLOT_WORD0, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h->tick_dev) h->tick = k.tick;
    std::memcpy(blob, &k, sizeof(k));
    if (h->P.task == RSX_TASK_VSS_V0) {
        // The task scalar of VSS-v0 (previous ball potential, vss_gym.py:256-283) is a function of the ball position
        // the next step starts from; the one-lane-per-env kernel recomputes it instead of keeping the row up to date.
        // The blob always carries the value, so that it restores into either kernel layout.  Same float expression
        // as the kernels' (this file is built with -ffp-contract=off; sqrtf is correctly rounded on both sides).
        const size_t B = (size_t)h->P.num_envs;
        const float* st = reinterpret_cast<const float*>((const char*)blob + sizeof(CkptHeader));
        float* aux = reinterpret_cast<float*>((char*)blob + sizeof(CkptHeader) + k.state_bytes);
        for (size_t e = 0; e < B; ++e) {
            const float bx = st[e], by = st[B + e];
            const float dx_d = (h->P.hl_goal + bx) * 100.0f, dx_a = (h->P.hl_goal - bx) * 100.0f, dy = by * 100.0f;
            const float dy2 = 2.0f * (dy * dy);
            const float dist_1 = -std::sqrt(dx_a * dx_a + dy2), dist_2 = std::sqrt(dx_d * dx_d + dy2);
            aux[(size_t)ROW_PREV_POT * B + e] = ((dist_1 + dist_2) * h->P.inv_len_cm - 1.0f) * 0.5f;
        }
    }
    return RSX_OK;
}

int rsx_task_checkpoint_load(rsx_sim* h, const void* blob, size_t bytes, void* stream) {
    RSX_ENTER_TASK(h);
    if (!blob || bytes < sizeof(CkptHeader)) return fail(RSX_ERR_ARG, "blob is null or truncated");
    CkptHeader k;
    std::memcpy(&k, blob, sizeof(k));
    CkptHeader want = ckpt_header(h);
    if (k.magic != CKPT_MAGIC || k.abi != want.abi) return fail(RSX_ERR_ARG, "not a checkpoint of this library version");
    if (k.model != want.model) return fail(RSX_ERR_ARG, "the checkpoint was taken under another version of the physics model (RSX_PHYSICS_MODEL)");
    if (k.kind != want.kind || k.task != want.task || k.n_blue != want.n_blue || k.n_yellow != want.n_yellow ||
        k.num_envs != want.num_envs || k.state_rows != want.state_rows || k.aux_rows != want.aux_rows || k.obs_dim != want.obs_dim)
        return fail(RSX_ERR_ARG, "the checkpoint was taken from a different configuration (simulator kind, team sizes, batch or task)");
    if (k.field_type != want.field_type || k.time_step_ms != want.time_step_ms || k.field_rows != want.field_rows)
        return fail(RSX_ERR_ARG, "the checkpoint was taken with another field type or time step");
    if (k.max_steps != want.max_steps)
        return fail(RSX_ERR_ARG, "the checkpoint was taken with another max_episode_steps (TimeLimit)");
    if (k.key0 != want.key0 || k.key1 != want.key1 || k.env_id_base != want.env_id_base)
        return fail(RSX_ERR_ARG, "the checkpoint was taken with another seed or env_id_base: attach the task with the same ones");
    // the section sizes follow from the configuration checked above; they are used for pointer arithmetic and as copy lengths below,
    // so a header that disagrees (a damaged file) is refused instead of being trusted
    if (k.state_bytes != want.state_bytes || k.aux_bytes != want.aux_bytes || k.obs_bytes != want.obs_bytes || k.flag_bytes != want.flag_bytes)
        return fail(RSX_ERR_ARG, "the checkpoint header is damaged (section sizes do not match its configuration)");
    if (bytes < ckpt_size(k)) return fail(RSX_ERR_ARG, "blob is truncated");
    hipStream_t s = (hipStream_t)stream;
    const char* p = (const char*)blob + sizeof(CkptHeader);
    h->host_state_valid = false;
    const size_t rowb = (size_t)h->P.num_envs * sizeof(float), pitch = (size_t)h->P.row_stride * sizeof(float);
    HIP_TRY(hipMemcpy2DAsync(h->d_state, pitch, p, rowb, rowb, (size_t)k.state_rows, hipMemcpyHostToDevice, s)); p += k.state_bytes;
    HIP_TRY(hipMemcpy2DAsync(h->d_aux, pitch, p, rowb, rowb, (size_t)k.aux_rows, hipMemcpyHostToDevice, s)); p += k.aux_bytes;
    HIP_TRY(hipMemcpyAsync(h->d_obs, p, k.obs_bytes, hipMemcpyHostToDevice, s)); p += k.obs_bytes;
    HIP_TRY(hipMemcpyAsync(h->d_final_obs, p, k.obs_bytes, hipMemcpyHostToDevice, s)); p += k.obs_bytes;
    HIP_TRY(hipMemcpyAsync(h->d_flags, p, k.flag_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(h->d_mslots, 0, (size_t)MSLOTS * RSX_METRICS * sizeof(unsigned long long), s));
    HIP_TRY(hipMemcpyAsync(h->d_metrics, k.metrics, sizeof(k.metrics), hipMemcpyHostToDevice, s));
    if (h->tick_dev) {   // device-keyed handle: every slot takes the blob's step counter; a refused-launch mark is cleared with it
        tick_fill(h, 0, h->tick_slots_alloc, k.tick, 0, s);
        HIP_TRY(launch_status());
        HIP_TRY(hipMemsetAsync(tick_words(h) + TICK_ERR_WORD, 0, sizeof(uint32_t), s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    h->tick = k.tick;
    h->task_ready = true;
    return RSX_OK;
}

}  // extern "C"

extern "C" {
int rsx_metrics_fold(rsx_sim* h, void* stream) {
    RSX_ENTER_TASK(h);
    rsx_launch(fold_metrics_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, h->d_metrics, h->d_mslots);
    HIP_TRY(launch_status());
    return RSX_OK;
}

int rsx_read_metrics(rsx_sim* h, int64_t out[RSX_METRICS], void* stream) {
    RSX_ENTER_TASK(h);
    if (!out) return fail(RSX_ERR_ARG, "out is null");
    hipStream_t s = (hipStream_t)stream;
    rsx_launch(fold_metrics_kernel, dim3(1), dim3(64), 0, s, h->d_metrics, h->d_mslots);
    HIP_TRY(launch_status());
    HIP_TRY(hipMemcpyAsync(out, h->d_metrics, RSX_METRICS * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    uint32_t refused = 0;
    if (h->tick_dev) HIP_TRY(hipMemcpyAsync(&refused, tick_words(h) + TICK_ERR_WORD, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (refused) return fail(RSX_ERR_STATE, "step counter exhausted: stepping launches of this device-keyed handle were refused on the device (out[] is valid; a handle takes at most 2^32 - 1 fused steps)");
    return RSX_OK;
}

}  // extern "C"
