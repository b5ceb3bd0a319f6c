"""ctypes binding of librsx_hip.so — the C-ABI declared in include/rsx.h.

This is the binding the reference would hold in place of ``import robosim``
(rsoccer_gym/Simulators/rsim.py:2).  It fails loudly when the HIP library is missing or no
GPU is visible: the product has no CPU path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSX_LIB") or os.path.join(_HERE, "librsx_hip.so")  # RSX_LIB: development builds

KIND_VSS, KIND_SSL = 0, 1
TASK_NONE, TASK_VSS_V0, TASK_SSL_STATIC_DEFENDERS = 0, 1, 2
TASK_SSL_DRIBBLING, TASK_SSL_CONTESTED, TASK_SSL_PASS_ENDURANCE = 3, 4, 5
TASK_SSL_SCRIMMAGE, TASK_SSL_SCRIMMAGE_CROWDED = 6, 7
FIELD_KEYS = (
    "length", "width", "penalty_length", "penalty_width", "goal_width", "goal_depth",
    "ball_radius", "rbt_distance_center_kicker", "rbt_kicker_thickness", "rbt_kicker_width",
    "rbt_wheel0_angle", "rbt_wheel1_angle", "rbt_wheel2_angle", "rbt_wheel3_angle",
    "rbt_radius", "rbt_wheel_radius", "rbt_motor_max_rpm",
)  # Entities/Field.py:5-21
N_METRICS = 8
X_ROWS = 2   # internal state rows behind get_state(): ball vertical velocity, ball spin (RSX_STATE_EXTRA_ROWS)
METRIC_NAMES = ("env_steps", "episodes", "goals_for", "goals_against", "return_sum_q20",
                "episode_len_sum", "truncated_episodes", "reserved")

# every symbol include/rsx.h declares (tests check the library exports each one)
SYMBOLS = (
    "rsx_abi_version", "rsx_last_error", "rsx_device_count", "rsx_create", "rsx_destroy",
    "rsx_get_field_params", "rsx_reset", "rsx_step", "rsx_get_state", "rsx_step_state", "rsx_wire_buffers", "rsx_step_wire", "rsx_set_state",
    "rsx_get_state_full", "rsx_dev_view_get", "rsx_step_dev", "rsx_step_dev_random", "rsx_step_dev_flip", "rsx_state_buffers",
    "rsx_reset_dev", "rsx_task_attach",
    "rsx_task_view_get", "rsx_task_reseed", "rsx_task_layout", "rsx_task_placement_cache_stats", "rsx_task_reset", "rsx_task_reset_to", "rsx_task_step",
    "rsx_task_step_n", "rsx_task_rollout", "rsx_read_metrics", "rsx_metrics_fold", "rsx_check_finite",
    "rsx_task_checkpoint_size", "rsx_task_checkpoint_save", "rsx_task_checkpoint_load",
    "rsx_task_enable_capture", "rsx_task_tick", "rsx_drop_pending_hip_error",
)


class RsxError(RuntimeError):
    pass


class DevView(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("n_robots", C.c_int32), ("state_dim", C.c_int32),
                ("cmd_dim", C.c_int32), ("state", C.c_void_p), ("cmds", C.c_void_p), ("row_stride", C.c_int32)]


class TaskView(C.Structure):
    _fields_ = [("task", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32),
                ("info_dim", C.c_int32), ("max_episode_steps", C.c_int32),
                ("obs", C.c_void_p), ("reward", C.c_void_p), ("terminated", C.c_void_p),
                ("truncated", C.c_void_p), ("info", C.c_void_p), ("final_obs", C.c_void_p),
                ("steps", C.c_void_p), ("actions", C.c_void_p), ("metrics", C.c_void_p), ("row_stride", C.c_int32)]


_lib = None


def load():
    """Load librsx_hip.so (once).  torch is imported first so that the HIP runtime already in
    the process (torch ships its own libamdhip64.so.7) is the one the library binds to —
    device pointers and streams are then shared with torch tensors."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RsxError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). rsoccer_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads libamdhip64 with the right SONAME first)
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.rsx_last_error.restype = C.c_char_p
    vp, ip, dp = C.c_void_p, C.c_int, C.POINTER(C.c_double)
    lib.rsx_create.argtypes = [C.POINTER(vp), ip, ip, ip, ip, ip, ip, ip]
    lib.rsx_destroy.argtypes = [vp]
    lib.rsx_get_field_params.argtypes = [vp, dp]
    lib.rsx_reset.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.rsx_step.argtypes = [vp, vp, vp]
    lib.rsx_get_state.argtypes = [vp, vp, vp]
    lib.rsx_step_state.argtypes = [vp, vp, vp, vp]
    lib.rsx_task_layout.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.rsx_task_placement_cache_stats.argtypes = [vp, C.POINTER(C.c_int64), vp]
    lib.rsx_set_state.argtypes = [vp, vp, vp]
    lib.rsx_wire_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.rsx_step_wire.argtypes = [vp, vp]
    lib.rsx_get_state_full.argtypes = [vp, vp, vp]
    lib.rsx_dev_view_get.argtypes = [vp, C.POINTER(DevView)]
    lib.rsx_step_dev.argtypes = [vp, vp]
    lib.rsx_step_dev_random.argtypes = [vp, ip, C.c_uint64, C.c_uint32, vp]
    lib.rsx_step_dev_flip.argtypes = [vp, vp]
    lib.rsx_state_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.rsx_reset_dev.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.rsx_task_attach.argtypes = [vp, ip, C.c_uint64, C.c_uint64, ip]
    lib.rsx_task_view_get.argtypes = [vp, C.POINTER(TaskView)]
    lib.rsx_task_reseed.argtypes = [vp, C.c_uint64, vp]
    lib.rsx_task_reset.argtypes = [vp, vp]
    lib.rsx_task_reset_to.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.rsx_task_step.argtypes = [vp, vp, vp]
    lib.rsx_task_step_n.argtypes = [vp, ip, vp]
    lib.rsx_task_rollout.argtypes = [vp, ip, vp]
    lib.rsx_read_metrics.argtypes = [vp, vp, vp]
    lib.rsx_metrics_fold.argtypes = [vp, vp]
    lib.rsx_task_checkpoint_size.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.rsx_task_checkpoint_save.argtypes = [vp, vp, C.c_size_t, vp]
    lib.rsx_task_checkpoint_load.argtypes = [vp, vp, C.c_size_t, vp]
    lib.rsx_check_finite.argtypes = [vp, C.POINTER(C.c_int64), vp]
    lib.rsx_task_enable_capture.argtypes = [vp, vp]
    lib.rsx_task_tick.argtypes = [vp, C.POINTER(C.c_uint32), vp]
    lib.rsx_drop_pending_hip_error.argtypes = []
    if lib.rsx_abi_version() != 6:
        raise RsxError("librsx_hip.so ABI version mismatch")
    _lib = lib
    return lib


def drop_pending_hip_error():
    """reads and clears the thread's pending HIP error (rsx_drop_pending_hip_error): what an aborted stream capture leaves behind"""
    return int(load().rsx_drop_pending_hip_error())


def _chk(rc):
    if rc != 0:
        raise RsxError(f"librsx_hip error {rc}: {load().rsx_last_error().decode()}")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, shape):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.size != int(np.prod(shape)):
        raise ValueError(f"expected {shape} values, got {a.shape}")
    return a.reshape(shape)


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can wrap library-owned memory."""

    def __init__(self, ptr, shape, typestr, owner, strides=None):
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2,
            "strides": None if strides is None else tuple(int(x) for x in strides),   # bytes
        }
        self._owner = owner  # keeps the handle alive while a tensor views its memory


class Sim:
    """A batch of ``num_envs`` simulator instances on one GPU (C handle ``rsx_sim``)."""

    def __init__(self, kind, field_type, n_blue, n_yellow, time_step_ms=25, num_envs=1,
                 device_id=0):
        self._lib = load()
        h = C.c_void_p()
        _chk(self._lib.rsx_create(C.byref(h), kind, field_type, n_blue, n_yellow,
                                  int(time_step_ms), int(num_envs), int(device_id)))
        self._h = h
        self.kind, self.field_type = kind, field_type
        self.n_blue, self.n_yellow = n_blue, n_yellow
        self.num_envs, self.device_id = int(num_envs), int(device_id)
        self.n_robots = n_blue + n_yellow
        self.cmd_dim = 2 if kind == KIND_VSS else 8              # rsim.py:92-101 / :137-153
        self.state_dim = 5 + (6 if kind == KIND_VSS else 11) * self.n_robots   # Entities/Frame.py:20-47,55-92
        self._view_cache = None   # raw device pointers are fetched on first use (see _view)
        self._tview = None
        self.task = TASK_NONE

    @property
    def _view(self):
        # asking for the raw pointers tells the library that the state can change behind its back
        # (it then stops serving rsx_get_state from the copy rsx_step brought home), so the
        # single-env robosim classes, which never touch device memory, do not ask
        if self._view_cache is None:
            v = DevView()
            _chk(self._lib.rsx_dev_view_get(self._h, C.byref(v)))
            assert (v.n_robots, v.state_dim, v.cmd_dim) == (self.n_robots, self.state_dim, self.cmd_dim)
            self._view_cache = v
        return self._view_cache

    # ---- lifetime ----
    def close(self):
        if getattr(self, "_h", None):
            self._lib.rsx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream(stream):
        if stream is None:
            return None
        return C.c_void_p(int(stream))

    # ---- robosim surface (host f64 wire format) ----
    def get_field_params(self):
        out = (C.c_double * 17)()
        _chk(self._lib.rsx_get_field_params(self._h, out))
        return dict(zip(FIELD_KEYS, [float(x) for x in out]))

    def reset(self, ball, blue, yellow, env_mask=None, stream=None):
        B = self.num_envs
        ball = _f64(ball, (B, 4))
        blue = _f64(blue, (B, self.n_blue, 3)) if self.n_blue else None
        yellow = _f64(yellow, (B, self.n_yellow, 3)) if self.n_yellow else None
        m = None if env_mask is None else np.ascontiguousarray(env_mask, dtype=np.uint8)
        _chk(self._lib.rsx_reset(self._h, _ptr(ball), _ptr(blue), _ptr(yellow), _ptr(m),
                                 self._stream(stream)))

    def step(self, cmds, stream=None):
        cmds = _f64(cmds, (self.num_envs, self.n_robots, self.cmd_dim))
        _chk(self._lib.rsx_step(self._h, _ptr(cmds), self._stream(stream)))

    def step_state(self, cmds, copy=True):
        """step(cmds) + get_state() in one crossing (rsx_step_state, null stream): returns the [B, state_dim] float64
        state.  ``cmds`` must be a C-contiguous float64 array of B * n_robots * cmd_dim values — the lean path of the
        robosim-shaped single-env objects (no conversions, no per-call stream object).  ``copy=False`` on a handle with
        wire buffers (more than 64 envs): the commands are copied into the pinned command buffer, the step runs through
        ``rsx_step_wire`` and the result is a VIEW of the pinned state buffer — valid until the next step, no pass over
        the 8 * B * state_dim bytes on the host."""
        if not copy:
            wire = self.wire_buffers()
            if wire is not None:
                np.copyto(wire[0], np.asarray(cmds).reshape(wire[0].shape))
                rc = self._lib.rsx_step_wire(self._h, None)
                if rc:
                    _chk(rc)
                return wire[1][:, :self.state_dim]
        out = np.empty((self.num_envs, self.state_dim), dtype=np.float64)
        rc = self._lib.rsx_step_state(self._h, cmds.ctypes.data, out.ctypes.data, None)
        if rc:
            _chk(rc)
        return out

    def wire_buffers(self):
        """(cmds, state): numpy views of the handle's pinned wire-format buffers (rsx_wire_buffers; batches of more than 64 envs) —
        cmds [B, n_robots, cmd_dim] float64 to be filled before ``step_wire()``, state [B, state_dim + 2] float64 (the
        ``get_state()`` vector + the two internal rows) valid after it.  None for handles without them."""
        if getattr(self, "_wire", None) is None:
            c, st = C.c_void_p(), C.c_void_p()
            if self._lib.rsx_wire_buffers(self._h, C.byref(c), C.byref(st)) != 0:
                self._wire = False
            else:
                nc, ns = self.num_envs * self.n_robots * self.cmd_dim, self.num_envs * (self.state_dim + X_ROWS)
                cm = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_double)), shape=(nc,)).reshape(self.num_envs, self.n_robots, self.cmd_dim)
                sv = np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_double)), shape=(ns,)).reshape(self.num_envs, self.state_dim + X_ROWS)
                self._wire = (cm, sv)
        return self._wire or None

    def step_wire(self, stream=None):
        """one step with the commands of ``wire_buffers()[0]``; the new state (all rows) lands in ``wire_buffers()[1]`` (rsx_step_wire)"""
        _chk(self._lib.rsx_step_wire(self._h, self._stream(stream)))

    def get_state(self, stream=None):
        out = np.empty((self.num_envs, self.state_dim), dtype=np.float64)
        _chk(self._lib.rsx_get_state(self._h, _ptr(out), self._stream(stream)))
        return out

    def get_state_full(self, stream=None):
        out = np.empty((self.num_envs, self.state_dim + X_ROWS), dtype=np.float64)
        _chk(self._lib.rsx_get_state_full(self._h, _ptr(out), self._stream(stream)))
        return out

    def set_state(self, state, stream=None):
        s = _f64(state, (self.num_envs, self.state_dim + X_ROWS))
        _chk(self._lib.rsx_set_state(self._h, _ptr(s), self._stream(stream)))

    # ---- device-resident path ----
    def step_dev(self, stream=None):
        _chk(self._lib.rsx_step_dev(self._h, self._stream(stream)))

    def step_dev_random(self, n=1, seed=0, first_tick=0, stream=None):
        """n steps with commands drawn on the device (Philox keyed by seed; ticks first_tick...)."""
        _chk(self._lib.rsx_step_dev_random(self._h, int(n), int(seed), int(first_tick), self._stream(stream)))

    def step_dev_flip(self, stream=None):
        """Double-buffered step_dev(): the two tensors of state_buffers() trade roles."""
        _chk(self._lib.rsx_step_dev_flip(self._h, self._stream(stream)))

    def state_buffers(self):
        """(current, other): two [state_dim+2, B] float32 views; after each step_dev_flip() the one
        that was current holds the previous frame."""
        cur, oth = C.c_void_p(), C.c_void_p()
        _chk(self._lib.rsx_state_buffers(self._h, C.byref(cur), C.byref(oth)))
        return self._rows(cur.value, self.state_dim + X_ROWS), self._rows(oth.value, self.state_dim + X_ROWS)

    def reset_dev(self, ball, blue, yellow, env_mask=None, stream=None):
        """reset() from device tensors: ball [B,4], blue [B,nb,3], yellow [B,ny,3] float32 CUDA,
        env_mask [B] uint8/bool CUDA or None.  Stream-ordered, no host copy."""
        import torch
        def prep(t, shape):
            if t is None:
                return None
            t = t.to(dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"expected {shape}, got {tuple(t.shape)}")
            return t
        B = self.num_envs
        ball = prep(ball, (B, 4))
        blue = prep(blue, (B, self.n_blue, 3)) if self.n_blue else None
        yellow = prep(yellow, (B, self.n_yellow, 3)) if self.n_yellow else None
        m = None
        if env_mask is not None:
            m = env_mask.to(dtype=torch.uint8).contiguous()
            if tuple(m.shape) != (B,):
                raise ValueError(f"env_mask must be [{B}]")
        ptr = lambda t: None if t is None else t.data_ptr()
        self._keep_reset = (ball, blue, yellow, m)   # alive until the launch has consumed them
        _chk(self._lib.rsx_reset_dev(self._h, ptr(ball), ptr(blue), ptr(yellow), ptr(m), self._stream(stream)))

    def _tensor(self, ptr, shape, typestr, strides=None):
        import torch
        return torch.as_tensor(_DevArray(ptr, shape, typestr, self, strides),
                               device=torch.device("cuda", self.device_id))

    def _rows(self, ptr, n_rows, rs=None):
        """[n_rows, B] float32 view of an SoA array whose rows are row_stride floats apart (dense unless the handle pads its rows:
        rsx.h, rsx_dev_view)"""
        rs = int(self._view.row_stride if rs is None else rs)
        return self._tensor(ptr, (n_rows, self.num_envs), "<f4", None if rs == self.num_envs else (4 * rs, 4))

    def state_tensor(self):
        """[state_dim+2, B] float32, zero-copy view of the SoA state."""
        return self._rows(self._view.state, self.state_dim + X_ROWS)

    def cmds_tensor(self):
        """[N*C, B] float32, zero-copy view of the SoA command buffer read by step_dev()."""
        return self._rows(self._view.cmds, self.n_robots * self.cmd_dim)

    # ---- fused tasks ----
    def task_attach(self, task, seed=0, env_id_base=0, max_episode_steps=0):
        _chk(self._lib.rsx_task_attach(self._h, task, seed, env_id_base, max_episode_steps))
        t = TaskView()
        _chk(self._lib.rsx_task_view_get(self._h, C.byref(t)))
        self._tview = t
        self.task = task
        self.obs_dim, self.act_dim, self.info_dim = t.obs_dim, t.act_dim, t.info_dim
        self.max_episode_steps = t.max_episode_steps

    def task_layout(self):
        """which tile layout steps this handle (rsx_task_layout): '8-lanes-per-env', 'one-lane-per-env', ..."""
        buf = C.create_string_buffer(64)
        _chk(self._lib.rsx_task_layout(self._h, buf, 64))
        return buf.value.decode()

    def placement_cache_stats(self, stream=None):
        """(resets served from the placement cache, resets placed inline), or (-1, -1) — rsx_task_placement_cache_stats"""
        out = (C.c_int64 * 2)()
        _chk(self._lib.rsx_task_placement_cache_stats(self._h, out, self._stream(stream)))
        return int(out[0]), int(out[1])

    def task_tensors(self):
        t, B = self._tview, self.num_envs
        return dict(
            obs=self._tensor(t.obs, (B, t.obs_dim), "<f4"),
            reward=self._tensor(t.reward, (B,), "<f4"),
            terminated=self._tensor(t.terminated, (B,), "|u1"),
            truncated=self._tensor(t.truncated, (B,), "|u1"),
            info=self._rows(t.info, t.info_dim, t.row_stride),
            final_obs=self._tensor(t.final_obs, (B, t.obs_dim), "<f4"),
            steps=self._tensor(t.steps, (B,), "<i4"),
            actions=self._tensor(t.actions, (B, t.act_dim), "<f4"),
            metrics=self._tensor(t.metrics, (N_METRICS,), "<i8"),
        )

    def task_reseed(self, seed, stream=None):
        """start over with another seed (rsx_task_reseed): the handle becomes what a fresh attach with that seed would be; reset next"""
        _chk(self._lib.rsx_task_reseed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, self._stream(stream)))

    def task_reset(self, stream=None):
        _chk(self._lib.rsx_task_reset(self._h, self._stream(stream)))

    def task_reset_to(self, ball, blue, yellow, env_mask=None, stream=None):
        B = self.num_envs
        ball = _f64(ball, (B, 4))
        blue = _f64(blue, (B, self.n_blue, 3)) if self.n_blue else None
        yellow = _f64(yellow, (B, self.n_yellow, 3)) if self.n_yellow else None
        m = None if env_mask is None else np.ascontiguousarray(env_mask, dtype=np.uint8)
        _chk(self._lib.rsx_task_reset_to(self._h, _ptr(ball), _ptr(blue), _ptr(yellow), _ptr(m),
                                         self._stream(stream)))

    def task_step(self, actions_ptr=None, stream=None):
        """actions_ptr: device address of a [B][act_dim] float32 array, or None = random.
        Hot path of the Python API: plain ints go straight to ctypes (argtypes are c_void_p)."""
        rc = self._lib.rsx_task_step(self._h, actions_ptr, stream)
        if rc:
            _chk(rc)

    def task_step_n(self, n, stream=None):
        _chk(self._lib.rsx_task_step_n(self._h, int(n), self._stream(stream)))

    def task_rollout(self, n, stream=None):
        _chk(self._lib.rsx_task_rollout(self._h, int(n), self._stream(stream)))

    def task_enable_capture(self, stream=None):
        """rsx_task_enable_capture: move the step counter to device memory so that stepping calls can be captured into a
        hipGraph (torch.cuda.CUDAGraph) and replayed.  Call once, outside any capture."""
        _chk(self._lib.rsx_task_enable_capture(self._h, self._stream(stream)))

    def task_tick(self, stream=None):
        """fused steps taken since attach (rsx_task_tick)"""
        n = C.c_uint32(0)
        _chk(self._lib.rsx_task_tick(self._h, C.byref(n), self._stream(stream)))
        return int(n.value)

    def check_finite(self, stream=None):
        """Number of non-finite floats in state / obs / reward / info (debugging aid; synchronises)."""
        n = C.c_int64(0)
        _chk(self._lib.rsx_check_finite(self._h, C.byref(n), self._stream(stream)))
        return int(n.value)

    def task_checkpoint(self, stream=None):
        """bytes of a checkpoint of the fused run (rsx_task_checkpoint_save): state, episode bookkeeping, noise
        state, step counter, metrics — see include/rsx.h"""
        n = C.c_size_t(0)
        _chk(self._lib.rsx_task_checkpoint_size(self._h, C.byref(n)))
        blob = np.empty(n.value, dtype=np.uint8)
        _chk(self._lib.rsx_task_checkpoint_save(self._h, blob.ctypes.data_as(C.c_void_p), n.value, self._stream(stream)))
        return blob

    def task_restore(self, blob, stream=None):
        """continue from ``task_checkpoint()`` (same simulator, batch, task, seed and env_id_base)"""
        blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, (bytes, bytearray)) else blob, dtype=np.uint8)
        _chk(self._lib.rsx_task_checkpoint_load(self._h, blob.ctypes.data_as(C.c_void_p), blob.size, self._stream(stream)))

    def metrics_fold(self, stream=None):
        """make the device copy of the episode counters (``task_tensors()["metrics"]``) exact, on ``stream``"""
        _chk(self._lib.rsx_metrics_fold(self._h, self._stream(stream)))

    def read_metrics(self, stream=None):
        out = np.zeros(N_METRICS, dtype=np.int64)
        _chk(self._lib.rsx_read_metrics(self._h, _ptr(out), self._stream(stream)))
        return out


def device_count():
    return int(load().rsx_device_count())
