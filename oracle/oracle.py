"""ctypes binding of the CPU oracle (oracle/librsx_oracle.so).

TEST INFRASTRUCTURE.  Importable only from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package rsoccer_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librsx_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = [os.path.join(_HERE, f) for f in ("rsx_oracle.c", "rsx_oracle_impl.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


_dp = C.POINTER(C.c_double)
X_ROWS = 2   # internal state entries behind get_state(): ball vertical velocity, ball spin (RSXO_XROWS)


def _d(a):
    return a.ctypes.data_as(_dp)


class OracleEnv:
    """One environment of the oracle; prec = 'f32' (the model the GPU implements bit-for-bit)
    or 'f64' (the reference's boundary precision)."""

    def __init__(self, kind, field_type, n_blue, n_yellow, time_step_ms=25, prec="f32"):
        self.L = lib()
        self.sfx = "_" + prec
        f = self._f("rsxo_create")
        f.restype = C.c_void_p
        self.h = f(kind, field_type, n_blue, n_yellow, time_step_ms)
        if not self.h:
            raise ValueError("oracle: bad configuration")
        self.h = C.c_void_p(self.h)
        self.kind, self.n_blue, self.n_yellow = kind, n_blue, n_yellow
        self.N = n_blue + n_yellow
        self.C = 2 if kind == 0 else 8
        self.state_dim = self._f("rsxo_state_dim")(self.h)
        self.obs_dim = self.act_dim = self.info_dim = 0

    def _f(self, name):
        return getattr(self.L, name + self.sfx)

    def close(self):
        if self.h:
            self._f("rsxo_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- robosim surface ----
    def field_params(self):
        out = np.zeros(17)
        self._f("rsxo_field_params")(self.h, _d(out))
        return out

    def reset(self, ball, blue, yellow):
        ball = np.ascontiguousarray(ball, dtype=np.float64).reshape(4)
        blue = np.ascontiguousarray(blue, dtype=np.float64).reshape(-1)
        yellow = np.ascontiguousarray(yellow, dtype=np.float64).reshape(-1)
        self._f("rsxo_reset")(self.h, _d(ball), _d(blue), _d(yellow))

    def step(self, cmds):
        cmds = np.ascontiguousarray(cmds, dtype=np.float64).reshape(self.N * self.C)
        self._f("rsxo_step")(self.h, _d(cmds))

    def step_random(self, seed, env_id, tick):
        """step() with commands drawn from Philox (mirror of rsx_step_dev_random)"""
        f = self._f("rsxo_step_random")
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32]
        f(self.h, int(seed), int(env_id), int(tick))

    def get_state(self):
        out = np.zeros(self.state_dim)
        self._f("rsxo_get_state")(self.h, _d(out))
        return out

    def get_state_full(self):
        out = np.zeros(self.state_dim + X_ROWS)
        self._f("rsxo_get_state_full")(self.h, _d(out))
        return out

    def set_state_full(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64).reshape(self.state_dim + X_ROWS)
        self._f("rsxo_set_state_full")(self.h, _d(s))

    # ---- tasks ----
    def task_attach(self, task, seed=0, env_id=0, max_steps=0):
        f = self._f("rsxo_task_attach")
        f.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int]
        if f(self.h, task, seed, env_id, max_steps):
            raise ValueError("oracle: task does not match the simulator")
        self.task = task
        self.obs_dim = {1: 4 + 7 * self.n_blue + 5 * self.n_yellow,
                        2: 4 + 8 * self.n_blue + 2 * self.n_yellow,
                        3: 5 + 8 * self.n_blue + 2 * self.n_yellow,
                        4: 4 + 8 * self.n_blue + 2 * self.n_yellow,
                        5: 4 + 6 * self.n_blue, 6: 2 + 2 * self.N, 7: 2 + 2 * self.N}[task]
        self.act_dim = {1: 2, 2: 5, 3: 4, 4: 5, 5: 3, 6: 4 * self.N, 7: 4 * self.N}[task]
        self.info_dim = {1: 6, 2: 8, 3: 1, 4: 9, 5: 2, 6: 2, 7: 2}[task]

    def task_reset(self):
        self._f("rsxo_task_reset")(self.h)

    def task_reset_to(self, ball, blue, yellow):
        ball = np.ascontiguousarray(ball, dtype=np.float64).reshape(4)
        blue = np.ascontiguousarray(blue, dtype=np.float64).reshape(-1)
        yellow = np.ascontiguousarray(yellow, dtype=np.float64).reshape(-1)
        self._f("rsxo_task_reset_to")(self.h, _d(ball), _d(blue), _d(yellow))

    def task_step(self, action=None):
        if action is None:
            self._f("rsxo_task_step")(self.h, None)
        else:
            a = np.ascontiguousarray(action, dtype=np.float32).reshape(self.act_dim)
            self._f("rsxo_task_step")(self.h, a.ctypes.data_as(C.POINTER(C.c_float)))

    def task_out(self):
        obs = np.zeros(self.obs_dim)
        fin = np.zeros(self.obs_dim)
        info = np.zeros(self.info_dim)
        rew = C.c_double()
        term = C.c_uint8()
        trunc = C.c_uint8()
        steps = C.c_int()
        met = np.zeros(8, dtype=np.int64)
        self._f("rsxo_task_out")(self.h, _d(obs), C.byref(rew), C.byref(term), C.byref(trunc),
                                 _d(info), _d(fin), C.byref(steps),
                                 met.ctypes.data_as(C.POINTER(C.c_int64)))
        return dict(obs=obs, reward=rew.value, terminated=term.value, truncated=trunc.value,
                    info=info, final_obs=fin, steps=steps.value, metrics=met)

    def task_last_cmds(self):
        out = np.zeros(self.N * self.C)
        self._f("rsxo_task_last_cmds")(self.h, _d(out))
        return out.reshape(self.N, self.C)

    # ---- pure functions (golden tests) ----
    def obs_eval(self):
        out = np.zeros(self.obs_dim)
        self._f("rsxo_task_obs_eval")(self.h, _d(out))
        return out

    def cmds_eval(self, act, theta_deg=0.0):
        act = np.ascontiguousarray(act, dtype=np.float64).reshape(-1)
        out = np.zeros(self.N * self.C)
        f = self._f("rsxo_task_cmds_eval")
        f.argtypes = [C.c_void_p, _dp, C.c_double, _dp]
        f(self.h, _d(act), float(theta_deg), _d(out))
        return out.reshape(self.N, self.C)

    def reward_eval(self, last_state, cmds, first_step):
        last = np.ascontiguousarray(last_state, dtype=np.float64)
        cmds = np.ascontiguousarray(cmds, dtype=np.float64).reshape(-1)
        rew = C.c_double()
        done = C.c_int()
        self._f("rsxo_task_reward_eval")(self.h, _d(last), _d(cmds), int(first_step),
                                         C.byref(rew), C.byref(done))
        return rew.value, bool(done.value)

    def ou_eval(self, x, normals):
        x = np.array(x, dtype=np.float64).reshape(-1)
        n = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1)
        self._f("rsxo_ou_eval")(self.h, _d(x), _d(n), len(x))
        return x

    def set_scalar(self, v):
        f = self._f("rsxo_task_set_scalar")
        f.argtypes = [C.c_void_p, C.c_double]
        f(self.h, float(v))

    def get_scalar(self):
        f = self._f("rsxo_task_get_scalar")
        f.restype = C.c_double
        return f(self.h)

    def norms(self):
        out = np.zeros(3)
        self._f("rsxo_task_norms")(self.h, _d(out))
        return out


def philox(ctr, key, rounds=10):
    """Philox4x32 with the published 10 rounds (known-answer vectors) or the engine's 7."""
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().rsxo_philox4x32(c, k, int(rounds), o)
    return list(o)


def sincos(a, prec="f32"):
    f = getattr(lib(), "rsxo_sincos_eval_" + prec)
    f.argtypes = [C.c_double, _dp, _dp]
    s, c = C.c_double(), C.c_double()
    f(float(a), C.byref(s), C.byref(c))
    return s.value, c.value


def log(x, prec="f32"):
    f = getattr(lib(), "rsxo_log_eval_" + prec)
    f.argtypes = [C.c_double]
    f.restype = C.c_double
    return f(float(x))


def set_threads(n):
    """Number of OpenMP threads used by vec_task_step."""
    lib().rsxo_set_threads(int(n))


def vec_task_step(envs, n_steps, prec="f32"):
    """OpenMP over envs (cpu_baseline)."""
    arr = (C.c_void_p * len(envs))(*[e.h for e in envs])
    getattr(lib(), "rsxo_vec_task_step_" + prec)(arr, len(envs), int(n_steps))
