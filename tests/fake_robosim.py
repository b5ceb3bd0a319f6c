"""CPU stand-in with the ``robosim`` surface, for CPU-only tests of the Python host layer.

Backed by the float64 instantiation of the oracle (tests only — the product never imports
this).  ``INJECT`` lets a test overwrite state entries after a given step, mirroring how the
golden episodes were scripted (tests/golden/make_golden.py)."""
import numpy as np

from oracle import oracle as O

INJECT = {}      # step index -> {state index: value}
TICK = {"t": 0}
PREC = "f64"


def arm(inject=None):
    INJECT.clear()
    INJECT.update(inject or {})
    TICK["t"] = 0


class _Sim:
    KIND = None

    def __init__(self, field_type, n_blue, n_yellow, time_step_ms, ball_pos, blue_pos, yellow_pos):
        self.o = O.OracleEnv(self.KIND, field_type, n_blue, n_yellow, time_step_ms, PREC)
        self.reset(np.asarray(ball_pos, float), np.asarray(blue_pos, float), np.asarray(yellow_pos, float))

    def get_field_params(self):
        from rsoccer_amd._lib import FIELD_KEYS
        return dict(zip(FIELD_KEYS, [float(v) for v in self.o.field_params()]))

    def get_state(self):
        return self.o.get_state()

    def step(self, cmds):
        self.o.step(cmds)
        t = TICK["t"]
        if t in INJECT:
            full = self.o.get_state_full()
            for idx, val in INJECT[t].items():
                full[idx] = val
            self.o.set_state_full(full)
        TICK["t"] = t + 1

    def reset(self, ball, blue, yellow):
        self.o.reset(ball, np.asarray(blue, float).reshape(-1), np.asarray(yellow, float).reshape(-1))


class VSS(_Sim):
    KIND = 0


class SSL(_Sim):
    KIND = 1
