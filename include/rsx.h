/*
 * rsx.h — C-ABI of the MI355X-native vectorised rSoccer step engine (librsx_hip.so).
 *
 * This is the drop-in boundary for the hot path of robocin/rSoccer: it takes the place of the
 * third-party `robosim` module (rc-robosim / rSim) that the reference binds in
 * rsoccer_gym/Simulators/rsim.py.  Every entry point cites the reference call site it replaces.
 * Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = error (text via rsx_last_error(), thread-local);
 *     no exception crosses the ABI.
 *   - "host" pointers are ordinary CPU memory owned by the caller (the reference wire format:
 *     float64, C-contiguous, rows = blue ids then yellow ids — rsim.py:92-101,129-153).
 *   - "dev" pointers are HIP device memory owned by the handle (valid until rsx_destroy); all
 *     device work is stream-ordered on the hipStream_t passed as `void* stream` (NULL = the
 *     null stream) and never synchronises implicitly, except the host-format calls
 *     (rsx_step / rsx_step_state / rsx_step_wire / rsx_get_state / rsx_get_state_full / rsx_reset / rsx_set_state /
 *     rsx_task_reset_to / rsx_read_metrics), which take or return host arrays and therefore
 *     synchronise that stream; rsx_create and rsx_task_attach synchronise the device once
 *     (their buffers are initialised before any caller stream can touch them).
 *   - no call changes the calling thread's current HIP device: the handle's device is made
 *     current for the duration of the call and the previous one is restored on return.
 *   - a caller compiled against this header checks rsx_abi_version() == RSX_ABI_VERSION BEFORE any other call: rsx_dev_view_get /
 *     rsx_task_view_get fill the caller's struct in the LIBRARY's layout (ABI 5 appended row_stride to both views — a caller built
 *     against an older header would have its stack overwritten), and every entry point assumes this header's argument lists.
 *   - rows of the device arrays are dense (row_stride == num_envs) below 786 432 envs; consumers that index base + f * num_envs on
 *     larger handles either use row_stride or set RSX_ROW_PAD=0 in the environment before rsx_create (dense rows at every batch,
 *     at the measured cost of DESIGN.md 3).
 *   - a handle is not thread-safe; distinct handles are independent.
 *   - there is NO CPU fallback: creation fails (RSX_ERR_NO_DEVICE) without a gfx950 device.
 *
 * Precision: the engine computes in float32.  Stated tolerance against the float64 instantiation
 * of the same model (the reference's boundary type, rsim.py:105), from the same state under the same
 * commands over 1 s (40 steps): |position| <= 1e-4 m, |velocity| <= 1e-3 m/s for every body
 * (tests/test_model_tolerance.py; crowded 22-robot scrums: 1e-2 m / 0.1 m/s, >= 95 % of the envs
 * within the tight bound).  The physics is this project's own 2-D model: rc-robosim's
 * sources are not part of the reference tree, so trajectories are not comparable with rSim's.
 *
 * Units (Entities/Frame.py:8): m, m/s, degrees, degrees/s.  VSS commands are wheel rad/s
 * (vss_gym.py:250-252); SSL commands are robot-local m/s and rad/s or wheel rad/s
 * (rsim.py:137-153).
 */
#ifndef RSX_H
#define RSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSX_ABI_VERSION 6
/* version of the 2-D step model the library implements (docs/PHYSICS.md): 1 = rounds 1-4; 2 = wall-aware contacts —
 * the SSL class since ABI 5 (probed shares, goal posts), the VSS class since ABI 6 (held axes, chord posts).  Stored in checkpoints:
 * a blob saved under another model version is refused (its trajectories would not continue bit-identically). */
#define RSX_PHYSICS_MODEL 2

/* kind: which robosim class the handle stands for (rsim.py:116 robosim.VSS, :169 robosim.SSL) */
#define RSX_KIND_VSS 0
#define RSX_KIND_SSL 1

/* fused task epilogues (obs / reward / done / OU noise / auto-reset computed on device) */
#define RSX_TASK_NONE                 0 /* raw simulator only                                         */
#define RSX_TASK_VSS_V0               1 /* rsoccer_gym/vss/env_vss/vss_gym.py:13  (obs 40, act 2)     */
#define RSX_TASK_SSL_STATIC_DEFENDERS 2 /* ssl/ssl_hw_challenge/static_defenders.py:12 (obs 24, act 5)*/
#define RSX_TASK_SSL_DRIBBLING        3 /* ssl/ssl_hw_challenge/dribbling.py:11         (obs 21, act 4)*/
#define RSX_TASK_SSL_CONTESTED        4 /* ssl/ssl_hw_challenge/contested_possession.py:11 (obs 14, act 5)*/
#define RSX_TASK_SSL_PASS_ENDURANCE   5 /* ssl/ssl_hw_challenge/pass_endurance.py:11    (obs 16, act 3)*/
/* synthetic SSL task in the style of the reference's example env (README.md:78-110) for team sizes no
 * registered id covers (BASELINE.json configs[3]: 11v11 on the division-A field): EVERY robot is
 * commanded — action [N][4] = v_x, v_y (robot-local, x 2.5 m/s), v_theta (x 10 rad/s), kick (5 m/s
 * when > 0.9) — obs = normalised ball and robot positions (2 + 2N), reward +1 / -1 and done on a goal
 * for blue / yellow.  Line-up: jittered 6 x 4 grid over the field (SCRIMMAGE) or packed around the ball,
 * the worst case for the all-pairs contact sweep (SCRIMMAGE_CROWDED). */
#define RSX_TASK_SSL_SCRIMMAGE         6
#define RSX_TASK_SSL_SCRIMMAGE_CROWDED 7

/* error codes */
#define RSX_OK              0
#define RSX_ERR_ARG        -1
#define RSX_ERR_NO_DEVICE  -2
#define RSX_ERR_HIP        -3
#define RSX_ERR_STATE      -4

/* number of entries of get_field_params(), in the order of Entities/Field.py:5-21 */
#define RSX_FIELD_PARAMS 17
/* internal state rows that follow the get_state() rows: ball vertical velocity (m/s), ball spin
 * about the vertical axis (rad/s) */
#define RSX_STATE_EXTRA_ROWS 2
/* metrics vector length (int64 each; see rsx_read_metrics) */
#define RSX_METRICS 8

typedef struct rsx_sim rsx_sim; /* opaque */

/* Device-side views, zero-copy.  SoA: row f of an [F][B] array is the contiguous run of B floats at
 * base + f*row_stride (one float per env).  row_stride >= num_envs: handles of 786 432 envs and more pad their rows
 * (64 KB + 256 B, 256 KB + 256 B from 1 572 864 envs) so that the rows of an env do not all sit at the same address modulo a large power of two (DRAM banks;
 * ABI 5).  Treat the arrays as strided 2-D views (torch: as_strided); a row by itself is dense.  state rows 0..state_dim-1 are exactly the reference's
 * get_state() layout (Entities/Frame.py:20-47 VSS, :55-92 SSL) transposed; rows state_dim and
 * state_dim + 1 hold the ball's vertical velocity and its spin (internal, needed to checkpoint
 * a chipped / spinning ball). */
typedef struct rsx_dev_view {
    int32_t num_envs;    /* B                                                                */
    int32_t n_robots;    /* N = n_blue + n_yellow                                            */
    int32_t state_dim;   /* 5 + 6N (VSS) | 5 + 11N (SSL)                                     */
    int32_t cmd_dim;     /* C: 2 (VSS) | 8 (SSL)  — per robot                                */
    float*  state;       /* [state_dim + 2][B] f32 SoA                                       */
    float*  cmds;        /* [N*C][B] f32 SoA, row = robot*C + col; read by rsx_step_dev      */
    int32_t row_stride;  /* floats from one row of state / cmds to the next (>= num_envs)    */
} rsx_dev_view;

typedef struct rsx_task_view {
    int32_t  task;
    int32_t  obs_dim;       /* 40 | 24 | 21 | 14 | 16 | 2 + 2N                               */
    int32_t  act_dim;       /* 2 | 5 | 4 | 5 | 3 | 4N                                        */
    int32_t  info_dim;      /* 6 | 8 | 1 | 9 | 2 | 2 : cumulative reward-shaping terms, order of the
                               reference's reward_shaping_total dict (dribbling, which has
                               none: its checkpoint counter)                                 */
    int32_t  max_episode_steps;
    float*   obs;           /* [B][obs_dim] f32 row-major (what a policy consumes)           */
    float*   reward;        /* [B] f32                                                       */
    uint8_t* terminated;    /* [B] u8 — task `done` of the step just taken                   */
    uint8_t* truncated;     /* [B] u8 — TimeLimit hit (rsoccer_gym/__init__.py:4,11)         */
    float*   info;          /* [info_dim][B] f32 SoA, values AFTER the step, BEFORE any
                               auto-reset clears them                                        */
    float*   final_obs;     /* [B][obs_dim] f32: terminal observation, written only for envs
                               whose episode ended in this step                              */
    int32_t* steps;         /* [B] i32: steps taken in the current episode                   */
    float*   actions;       /* [B][act_dim] f32: staging buffer callers may fill and pass to
                               rsx_task_step (any device pointer of that shape works)        */
    int64_t* metrics;       /* [RSX_METRICS] i64 device counters, maintained by the step kernels
                               (stream-ordered).  [0] is exact after every launch; the episode
                               counters [1..6] are exact after rsx_metrics_fold /
                               rsx_read_metrics (the kernels add into per-block partial sums) */
    int32_t  row_stride;    /* floats from one row of `info` to the next (>= num_envs; the same value
                               as rsx_dev_view.row_stride)                                   */
} rsx_task_view;

/* ---- diagnostics ---------------------------------------------------------------------- */
int         rsx_abi_version(void);
const char* rsx_last_error(void);
/* number of visible HIP devices (0 when none); never fails */
int         rsx_device_count(void);

/* ---- robosim.VSS / robosim.SSL replacement (batched) ------------------------------------ */

/* ctor — replaces robosim.VSS(...) rsim.py:116-124 and robosim.SSL(...) rsim.py:169-177.
 * field_type: VSS 0 = 3v3, 1 = 5v5; SSL 0 = div-B, 1 = div-A 11v11, 2 = hardware-challenge.
 * num_envs independent copies live on device `device_id`.  Initial poses are the adapter's
 * dummy line-up (rsim.py:20-24): ball at origin, blue at x = -0.2*(i+1), yellow x = +0.2*(i+1). */
int rsx_create(rsx_sim** out, int kind, int field_type, int n_blue, int n_yellow,
               int time_step_ms, int num_envs, int device_id);
/* destructor — replaces `del self.simulator` rsim.py:41 */
int rsx_destroy(rsx_sim* h);

/* get_field_params() — rsim.py:50; out[17] in the order of Entities/Field.py:5-21 */
int rsx_get_field_params(const rsx_sim* h, double out[RSX_FIELD_PARAMS]);

/* reset(ball, blue, yellow) — rsim.py:38,52-75.  Host f64: ball [B][4] = x,y,vx,vy;
 * blue [B][n_blue][3], yellow [B][n_yellow][3] = x,y,theta(deg) (NULL allowed when that team
 * is empty).  env_mask [B] u8 or NULL (= all): only envs with a non-zero mask are teleported.
 * Robot velocities, wheel speeds, infrared and ball height are zeroed. */
int rsx_reset(rsx_sim* h, const double* ball, const double* blue, const double* yellow,
              const uint8_t* env_mask, void* stream);

/* step(cmds) — rsim.py:102 (VSS, [B][N][2]) and rsim.py:155 (SSL, [B][N][8]); host f64.
 * The device-side command buffer (rsx_dev_view.cmds, what rsx_step_dev reads) is UNSPECIFIED after this call: handles of at
 * most 64 envs read the commands straight from pinned host memory and never write them there.  Callers that mix the host-format
 * and the device-resident calls fill view.cmds themselves before rsx_step_dev (results of rsx_step itself do not depend on the
 * path: tests/test_gpu_parity.py::test_host_format_step_with_and_without_zero_copy). */
int rsx_step(rsx_sim* h, const double* cmds, void* stream);

/* get_state() — rsim.py:105,158; out [B][state_dim] host f64 */
int rsx_get_state(rsx_sim* h, double* out, void* stream);

/* step(cmds) followed by get_state() — rsim.py:102,105 / :155,158 are always called as a pair (RSim*.send_commands,
 * RSim*.get_frame): one FFI crossing instead of two.  Handles of at most 64 envs (the robosim-shaped single-env
 * objects) run rsx_step / rsx_step_state without any copy: the kernel reads the commands from, and mirrors the new
 * state into, pinned host memory — one launch and one synchronisation per step. */
int rsx_step_state(rsx_sim* h, const double* cmds, double* state_out, void* stream);

/* The same pair for batches of more than 64 envs WITHOUT any pass of a CPU thread over the data (ABI 6).  Such handles own two pinned
 * host buffers in the reference's wire format — commands [B][N][C] float64 (what rsim.py:92-101 / :129-153 build), state
 * [B][state_dim + RSX_STATE_EXTRA_ROWS] float64 (the get_state() vector of rsim.py:105,158 followed by the two internal rows) — and
 * convert between them and the device's float32 row layout ON THE DEVICE: small kernels read / write the pinned buffers across PCIe
 * around the step kernel; one synchronisation.  rsx_wire_buffers returns the two buffers (valid until rsx_destroy; either pointer
 * argument may be NULL); the caller writes commands into *cmds, calls rsx_step_wire, and reads the new state from *state.
 * rsx_step / rsx_get_state / rsx_step_state on such handles are one memcpy in front of / behind the same path (before ABI 6: a
 * transposing float64 <-> float32 loop on the calling thread plus two staging copies — 192 us per step + state at 4096 envs).
 * Handles of at most 64 envs have no wire buffers (RSX_ERR_STATE): their rsx_step_state is already copy-free. */
int rsx_wire_buffers(rsx_sim* h, double** cmds, double** state);
int rsx_step_wire(rsx_sim* h, void* stream);

/* full-state restore (checkpoint/resume, also used by parity tests): state
 * [B][state_dim + RSX_STATE_EXTRA_ROWS] host f64 = get_state() layout + ball vertical velocity
 * + ball spin. rsx_get_state_full is its inverse.  (The infrared entry of an SSL robot is a flag, Entities/Frame.py:86: give 0 or 1.
 * Any other non-zero value reads as "on"; a step WITH physics rewrites the entry as 0 / 1, a step of a handle with time_step_ms 0
 * leaves what it finds when it steps in place and writes 1 when it writes another buffer, rsx_step_dev_flip.)
 * On a handle with a task attached this overwrites the SIMULATOR state only: observations, episode bookkeeping and the per-episode
 * task scalars (previous ball potential, checkpoint / stalled-step counters) are left as the last step wrote them, so the shaping
 * terms of the next step's reward refer to a frame that no longer exists — and how they do differs between kernel layouts (the
 * one-lane-per-env VSS kernel derives the previous potential from the ball position it finds).  To move envs of a fused run use
 * rsx_task_reset_to (new episode) or rsx_task_checkpoint_load (everything). */
int rsx_set_state(rsx_sim* h, const double* state, void* stream);
int rsx_get_state_full(rsx_sim* h, double* out, void* stream);

/* ---- device-resident path (no host copies) -------------------------------------------- */
int rsx_dev_view_get(rsx_sim* h, rsx_dev_view* out);
/* advance all envs by time_step_ms using the commands currently in view.cmds */
int rsx_step_dev(rsx_sim* h, void* stream);

/* The same step, double-buffered: the new state is written to the handle's second state buffer,
 * which then becomes the current one — the buffer that was current holds the PREVIOUS frame
 * (the reference's `last_frame`, vss_gym_base.py:80) without a copy.  rsx_state_buffers returns both
 * pointers (layout of rsx_dev_view.state); after every flip they trade places. */
int rsx_step_dev_flip(rsx_sim* h, void* stream);
int rsx_state_buffers(rsx_sim* h, float** current, float** other);
/* reset(ball, blue, yellow) of rsim.py:38 from DEVICE arrays (f32, same shapes as rsx_reset),
 * env_mask_dev [B] u8 device or NULL: stream-ordered, no host copy, no synchronisation. */
int rsx_reset_dev(rsx_sim* h, const float* ball_dev, const float* blue_dev, const float* yellow_dev,
                  const uint8_t* env_mask_dev, void* stream);

/* n steps with commands drawn on the device instead of read from view.cmds (benchmark / soak mode of
 * the raw simulator): robot k of env e takes Philox block
 * (e, first_tick + i, k, 4) keyed by `seed` in launch i — VSS: wheel speeds U(-1, 1) x the motor
 * limit; SSL: robot-local velocities U(-1, 1) x (2.5 m/s, 2.5 m/s, 10 rad/s). */
int rsx_step_dev_random(rsx_sim* h, int n, uint64_t seed, uint32_t first_tick, void* stream);

/* ---- fused task epilogues -------------------------------------------------------------- */

/* Attach a task to a handle whose kind / robot counts match it (VSS_V0: VSS, n_blue >= 1;
 * STATIC_DEFENDERS: SSL 1vN; DRIBBLING: SSL 1v4; CONTESTED: SSL 1v1; PASS_ENDURANCE: SSL 2v0;
 * SCRIMMAGE / SCRIMMAGE_CROWDED: SSL, any team sizes).  seed + (env_id_base + local env index) key every random draw,
 * so results do not depend on batch size, batch position or sharding.
 * max_episode_steps <= 0 selects the registry value (1200 / 1000 / 4800 / 1200 / 1200; scrimmage 1200). */
/* Random streams: placement draws are keyed by (seed, global env id, episode, index); the per-step draws
 * (random actions, OU noise) by (seed, global env id, number of fused steps the handle has taken since
 * attach) — a run is reproducible from its seed and its sequence of calls.
 * Limits (both are 32-bit words of the Philox counter, and both are checked, never wrapped):
 *   - env_id_base + num_envs <= 2^32, else RSX_ERR_ARG;
 *   - every [rows][num_envs] float array of a handle stays below 4 GB (rows x num_envs < 2^30: VSS 3v3 ~21 M envs,
 *     SSL 1v6 ~12 M, SSL 11v11 ~4 M), else RSX_ERR_ARG from rsx_create / rsx_task_attach: the kernels address a
 *     row with a 32-bit byte offset;
 *   - a handle takes at most 2^32 - 1 fused steps (rsx_task_step / _step_n / _rollout; about 11 h at 10^5 calls/s);
 *     the call that would exceed it returns RSX_ERR_STATE and changes nothing.  The counter is part of the checkpoint. */
int rsx_task_attach(rsx_sim* h, int task, uint64_t seed, uint64_t env_id_base,
                    int max_episode_steps);
int rsx_task_view_get(rsx_sim* h, rsx_task_view* out);
/* Start over with another seed (ABI 6): afterwards the handle is what a fresh rsx_task_attach(task, seed, same env_id_base, same
 * max_episode_steps) would have left — every random stream re-keyed, step counter 0, episode ids, per-env scalars, metrics and
 * placement cache cleared; the buffers (and every pointer of rsx_task_view) stay where they are; a handle switched by
 * rsx_task_enable_capture stays switched.  The next call must be rsx_task_reset / rsx_task_reset_to.  Stream-ordered; not
 * capturable (it changes host state).  What `reset(seed=...)` of a gymnasium-style vector env maps to (README.md:116-133 seeds
 * through `env.reset(seed=...)`; the reference's own tasks ignore it: SURVEY.md appendix D-1). */
int rsx_task_reseed(rsx_sim* h, uint64_t seed, void* stream);
/* Which tile layout steps this handle's envs (chosen at attach from task and batch size; results are identical in all):
 * "8-lanes-per-env" / "16-..." / "32-..." / "64-...", "32-lanes-per-env-large-batch", "one-lane-per-env",
 * "four-lanes-per-env".  NUL-terminated into out[n] — for profiles and benchmark lines, so that nothing outside the
 * library restates its thresholds. */
int rsx_task_layout(rsx_sim* h, char* out, size_t n);
/* Introspection of the placement cache (handles of STATIC_DEFENDERS 1v6 with at most 16 384 envs: every
 * single-step launch carries helper workgroups that compute each env's NEXT episode's random placement ahead of time —
 * a pure function of seed, global env id and episode — so that the wave that resets an env only copies it; results are
 * those of the inline placement, bit for bit).  out[0] = resets served from the cache, out[1] = placed inline; both -1
 * when the handle has no cache or the counters are off (set RSX_PCACHE_STATS=1 before rsx_task_attach; RSX_NO_PCACHE=1
 * disables the cache).  Synchronises `stream`. */
int rsx_task_placement_cache_stats(rsx_sim* h, int64_t out[2], void* stream);

/* reset(): new random placement for every env (vss_gym.py:194-233, static_defenders.py:214-254),
 * episode counters cleared, obs written. */
int rsx_task_reset(rsx_sim* h, void* stream);
/* reset() onto an explicit placement (same arrays as rsx_reset); obs written.  env_mask (host, num_envs bytes, or NULL = all): only
 * the envs with a non-zero byte are touched.
 * Both resets open a new episode for the envs they touch: observation written, step count 0, info rows and episode sums zero, noise
 * state cleared.  `reward`, `terminated` and `truncated` are outputs of step(): they keep what each env's LAST step wrote — for the
 * envs a mask leaves alone and for the re-placed ones alike — until that env's next step (reset() returns (obs, {}):
 * vss_gym_base.py:92-106). */
int rsx_task_reset_to(rsx_sim* h, const double* ball, const double* blue, const double* yellow,
                      const uint8_t* env_mask, void* stream);

/* The three stepping calls below return RSX_ERR_STATE until rsx_task_reset or
 * rsx_task_reset_to has opened the first episode.
 *
 * step(action): actions_dev [B][act_dim] f32 device memory, or NULL = uniform random actions
 * drawn on device (the "random actions" benchmark configuration).  One kernel launch does
 * action -> commands (+ OU noise for the non-agent robots), physics, observation, reward,
 * done, TimeLimit and same-step auto-reset. */
int rsx_task_step(rsx_sim* h, const float* actions_dev, void* stream);
/* n consecutive random-action steps = n kernel launches issued from C (no per-step FFI cost). */
int rsx_task_step_n(rsx_sim* h, int n, void* stream);
/* n consecutive random-action steps inside ONE launch (state stays in registers between
 * steps; obs / reward / done buffers hold the values of the last step).  Same results as n single steps; the
 * library may issue it that way where that is faster (SSL 11v11 handles of >= 49 152 envs do; the crowded line-up from 196 608). */
int rsx_task_rollout(rsx_sim* h, int n, void* stream);

/* ---- hipGraph / stream capture ------------------------------------------------------------------
 * A trainer steps once per policy action (vss_gym_base.py:72-90; the loop of the reference's README.md:116-133), and
 * with a small policy network that loop is bound by launch overheads: the usual cure is to capture
 * policy(obs) -> step(actions) into one hipGraph (torch.cuda.CUDAGraph) and replay it.
 *
 * By default the handle's step counter — the key of the per-step random draws (random actions, OU noise) and the
 * parity of the placement cache — is a HOST count baked into each launch's arguments.  A captured launch would replay
 * one tick for ever, so the three stepping calls REFUSE to be captured in that mode: RSX_ERR_STATE, nothing enqueued
 * (the capture itself stays valid).
 *
 * rsx_task_enable_capture(h, stream) moves the counter to device memory for the rest of the handle's life (one 32-bit
 * slot per workgroup behind the metrics vector: every workgroup of a stepping launch reads its slot and writes it back
 * advanced, no atomics, no extra launch).  Call it once, OUTSIDE any capture, stream-ordered after the handle's earlier
 * work.  From then on rsx_task_step / _step_n / _rollout (and rsx_task_reset) may be captured and replayed any number
 * of times, mixed freely with eager calls; every replay advances the counter exactly as the eager call would have, so a
 * run is bit-identical whether its steps were issued eagerly, captured, or both.  Costs nothing on handles that never
 * call it; on handles that did, a stepping launch reads one more dword.
 *   - the 2^32 - 1 step limit is then enforced on the device: a launch that would wrap the counter changes nothing and
 *     sets a mark that rsx_task_tick / rsx_read_metrics report as RSX_ERR_STATE;
 *   - rsx_task_checkpoint_save / _load carry the counter in either mode;
 *   - RSX_DEBUG_FINITE=1 (a synchronous scan) makes captured stepping calls fail with RSX_ERR_STATE;
 *   - calls that keep host-side state stay uncapturable and say so: rsx_step_dev_flip (buffer roles).  The raw
 *     rsx_step_dev / rsx_reset_dev launches hold no host state and can be captured as they are;
 *   - HIP's per-thread last-error slot: kernel launches of this library report through their own return values and never read or
 *     clear the slot (a caller's pending error stays the caller's) — with ONE exception: rsx_task_enable_capture clears it.  A
 *     capture that a stepping call refused is usually aborted by the caller's framework, and the aborted capture leaves
 *     `invalid argument` behind for the next capture to trip over (torch.cuda.graph does); the call that prepares the next
 *     attempt is where it is dropped. */
int rsx_task_enable_capture(rsx_sim* h, void* stream);
/* Reads AND clears the calling thread's pending HIP error; returns its hipError_t value (0 = none was pending).  For callers that
 * recover from an aborted capture without going through rsx_task_enable_capture (e.g. hook-written envs on the raw step, whose
 * rsx_step_dev_flip refused to be captured). */
int rsx_drop_pending_hip_error(void);
/* fused steps this handle has taken since attach (the counter above).  Device-keyed handles: synchronises `stream`. */
int rsx_task_tick(rsx_sim* h, uint32_t* out, void* stream);

/* Debugging aid: number of non-finite floats in the state rows and, with a task attached, in the
 * observations, rewards and info rows.  Synchronises `stream`.  With RSX_DEBUG_FINITE=1 in the
 * environment every stepping call (rsx_step_dev, rsx_task_step, rsx_task_step_n, rsx_task_rollout)
 * runs this scan afterwards and returns RSX_ERR_STATE when it finds one (the reference has no such
 * guard: e.g. rsoccer_gym/vss/env_vss/vss_gym.py:298 divides by a distance that can be zero). */
int rsx_check_finite(rsx_sim* h, int64_t* n_bad, void* stream);

/* ---- checkpoint / resume of a fused run ----------------------------------------------------
 * (the reference cannot: robosim exposes no way to restore velocities, rsim.py:52-75, and its tasks keep their
 * episode state — OU noise, step counters, cumulative reward terms — in Python attributes.)
 * The blob holds every per-env buffer of the handle (state incl. the two internal rows, step / episode counters,
 * OU noise, cumulative info terms, last observation / reward / flags), the handle's step counter (the key of the
 * per-step random draws) and the metrics.  Loading it into a handle created with the same simulator kind, team
 * sizes, batch size and attached with the same task, seed and env_id_base makes every following step bit-identical
 * to what the saving handle would have produced — across kernel layouts (RSX_LAYOUT) and processes.  Host blob,
 * both calls synchronise `stream`. */
int rsx_task_checkpoint_size(rsx_sim* h, size_t* bytes);
int rsx_task_checkpoint_save(rsx_sim* h, void* blob, size_t bytes, void* stream);
int rsx_task_checkpoint_load(rsx_sim* h, const void* blob, size_t bytes, void* stream);

/* metrics, int64[RSX_METRICS], accumulated on device since attach (payload of the multi-GPU
 * all-reduce): 0 env_steps, 1 episodes, 2 goals_for (blue), 3 goals_against (yellow),
 * 4 sum of episode returns in 2^-20 fixed point, 5 sum of episode lengths,
 * 6 truncated episodes, 7 reserved.  Synchronises `stream`. */
int rsx_read_metrics(rsx_sim* h, int64_t out[RSX_METRICS], void* stream);
/* Device-side readers of rsx_task_view.metrics (e.g. an RCCL all-reduce of the 64 bytes) call this
 * first: one tiny launch on `stream` that adds the step kernels' partial episode counters into
 * metrics[1..6].  (Atomics of a whole grid on one cache line serialise: at 10^6 envs they, not the
 * physics, set the step time.)  rsx_read_metrics does it itself. */
int rsx_metrics_fold(rsx_sim* h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RSX_H */
