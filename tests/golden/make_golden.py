"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (needs /root/reference; never runs on the GPU box, never at
test time):

    python tests/golden/make_golden.py

What it does: puts three throw-away stub packages on sys.path (gymnasium / pygame / robosim —
the reference's third-party imports, none of which is installed here), imports the reference's
own env classes from /root/reference, drives them, and records inputs and outputs of the
Python-side arithmetic this project restates: frame parsing, command packing, observation,
reward/done/info, normalisers, seeded placement, OU noise, KD-tree queries.

The stub `robosim` has no physics of its own: `step()` hands the commands to a callback, which
here advances this project's float64 oracle so that the recorded state sequences are physically
plausible (the reference's reward code only ever sees states, so any state sequence pins it).
Only data is written (npz); no reference source is copied.
"""
import importlib
import json
import os
import random
import sys
import tempfile
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

STUBS = {
    "gymnasium/__init__.py": """
        from . import spaces
        from .envs import registration
        class Env:
            metadata = {}
            def reset(self, *, seed=None, options=None):
                return None
    """,
    "gymnasium/spaces.py": """
        import numpy as np
        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.shape = tuple(shape); self.dtype = np.dtype(dtype)
                self.low = np.full(self.shape, low, dtype=self.dtype)
                self.high = np.full(self.shape, high, dtype=self.dtype)
    """,
    "gymnasium/envs/__init__.py": "from . import registration\n",
    "gymnasium/envs/registration.py": """
        REGISTRY = {}
        def register(id, entry_point=None, max_episode_steps=None, kwargs=None, **kw):
            REGISTRY[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, kwargs=kwargs or {})
    """,
    "pygame/__init__.py": "",
    "robosim/__init__.py": """
        import numpy as np
        FIELD = {}            # kind -> field dict (set by the generator)
        LOG = []              # (method, args) records
        class _Sim:
            KIND = None
            def __init__(self, *args):
                LOG.append(("ctor", self.KIND, args))
                self.args = args
                self.state = None
                self.on_step = None
                self.on_reset = None
            def get_field_params(self):
                return dict(FIELD[self.KIND])
            def get_state(self):
                return np.array(self.state, dtype=np.float64)
            def step(self, cmds):
                LOG.append(("step", np.array(cmds, copy=True)))
                if self.on_step is not None:
                    self.on_step(np.array(cmds, copy=True))
            def reset(self, ball, blue, yellow):
                LOG.append(("reset", np.array(ball, copy=True), np.array(blue, copy=True), np.array(yellow, copy=True)))
                if self.on_reset is not None:
                    self.on_reset(np.array(ball), np.array(blue), np.array(yellow))
        class VSS(_Sim):
            KIND = "vss"
        class SSL(_Sim):
            KIND = "ssl"
    """,
}

FIELD_KEYS = ("length", "width", "penalty_length", "penalty_width", "goal_width", "goal_depth",
              "ball_radius", "rbt_distance_center_kicker", "rbt_kicker_thickness", "rbt_kicker_width",
              "rbt_wheel0_angle", "rbt_wheel1_angle", "rbt_wheel2_angle", "rbt_wheel3_angle",
              "rbt_radius", "rbt_wheel_radius", "rbt_motor_max_rpm")


def frame_to_arrays(frame, nb, ny):
    ball = np.array([frame.ball.x, frame.ball.y, frame.ball.v_x, frame.ball.v_y])
    blue = np.array([[frame.robots_blue[i].x, frame.robots_blue[i].y, frame.robots_blue[i].theta] for i in range(nb)]).reshape(nb, 3)
    yel = np.array([[frame.robots_yellow[i].x, frame.robots_yellow[i].y, frame.robots_yellow[i].theta] for i in range(ny)]).reshape(ny, 3)
    return ball, blue, yel


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    from oracle import oracle as O
    O.build()
    tmp = tempfile.mkdtemp(prefix="rsx_stubs_")
    for rel, src in STUBS.items():
        p = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(textwrap.dedent(src))
    sys.path.insert(0, REF)
    sys.path.insert(0, tmp)
    import robosim
    # plain Python floats, like the dict a pybind11 module returns (numpy scalars would change the
    # reference's float32/float64 promotion)
    vss_field = dict(zip(FIELD_KEYS, (float(v) for v in O.OracleEnv(0, 0, 3, 3, 25, "f64").field_params())))
    ssl_field = dict(zip(FIELD_KEYS, (float(v) for v in O.OracleEnv(1, 2, 1, 6, 25, "f64").field_params())))
    robosim.FIELD["vss"] = vss_field
    robosim.FIELD["ssl"] = ssl_field

    from rsoccer_gym.vss.env_vss.vss_gym import VSSEnv
    from rsoccer_gym.ssl.ssl_hw_challenge.static_defenders import SSLHWStaticDefendersEnv
    from rsoccer_gym.Entities import FrameVSS, FrameSSL, Robot
    from rsoccer_gym.Simulators.rsim import RSimVSS, RSimSSL
    from rsoccer_gym.Utils import KDTree
    from rsoccer_gym.Utils.Utils import OrnsteinUhlenbeckAction
    import gymnasium

    rng = np.random.default_rng(20240928)
    out = {}

    # ------------------------------------------------------------------ VSS-v0
    env = VSSEnv()
    sim = env.rsim.simulator
    out["vss_field"] = np.array([vss_field[k] for k in FIELD_KEYS])
    out["vss_norms"] = np.array([env.max_pos, env.max_v, env.max_w])
    # (a) observations of arbitrary states
    K = 48
    states = np.zeros((K, 41))
    states[:, 0:2] = rng.uniform(-1.2, 1.2, (K, 2)); states[:, 2] = 0.0215
    states[:, 3:5] = rng.uniform(-2.5, 2.5, (K, 2))
    for k in range(6):
        o = 5 + 6 * k
        states[:, o:o + 2] = rng.uniform(-1.2, 1.2, (K, 2))
        states[:, o + 2] = rng.uniform(-400, 400, K)
        states[:, o + 3:o + 5] = rng.uniform(-2.0, 2.0, (K, 2))
        states[:, o + 5] = rng.uniform(-2500, 2500, K)
    obs = []
    for s in states:
        sim.state = s
        env.frame = env.rsim.get_frame()
        obs.append(env._frame_to_observations())
    out["vss_obs_states"] = states
    out["vss_obs"] = np.array(obs, dtype=np.float32)
    # (b) action -> wheel commands (incl. dead zone and clipping)
    acts = np.concatenate([rng.uniform(-1.3, 1.3, (40, 2)),
                           np.array([[0.05, -0.05], [0.03, 0.0], [0.0417, -0.0418], [1.0, -1.0], [0.5, -0.3]])])
    out["vss_wheel_actions"] = acts
    out["vss_wheel_cmds"] = np.array([env._actions_to_v_wheels(a) for a in acts], dtype=np.float64)
    # (c) OU process: normals drawn from the same global stream, then the reference's samples
    np.random.seed(0)
    normals = np.random.normal(size=(30, 2))
    np.random.seed(0)
    ou = OrnsteinUhlenbeckAction(env.action_space, dt=0.025)
    out["ou_normals"] = normals
    out["ou_samples"] = np.array([ou.sample() for _ in range(30)], dtype=np.float64)
    # (d) seeded placement
    seeds = [0, 1, 2, 7, 12345]
    pl = []
    for sd in seeds:
        random.seed(sd)
        fr = env._get_initial_positions_frame()
        b, bl, ye = frame_to_arrays(fr, 3, 3)
        pl.append(np.concatenate([b, bl.ravel(), ye.ravel()]))
    out["vss_place_seeds"] = np.array(seeds)
    out["vss_place"] = np.array(pl)
    # (e) whole episodes through the public reset()/step() path on oracle-f64 physics
    orc = O.OracleEnv(0, 0, 3, 3, 25, "f64")
    inject = {}
    tick = {"t": 0}

    def on_reset(ball, blue, yellow):
        orc.reset(ball, blue, yellow)
        sim.state = orc.get_state()

    def on_step(cmds):
        orc.step(cmds)
        st = orc.get_state()
        if tick["t"] in inject:
            st = st.copy()
            for idx, val in inject[tick["t"]].items():
                st[idx] = val
            full = orc.get_state_full(); full[:41] = st; orc.set_state_full(full)
        sim.state = st
        tick["t"] += 1

    sim.on_reset, sim.on_step = on_reset, on_step
    episodes = []
    for ep, (T, inj) in enumerate([(40, {39: {0: 0.76}}), (25, {24: {0: -0.751}}), (60, {})]):
        random.seed(100 + ep); np.random.seed(200 + ep)
        inject.clear(); inject.update(inj); tick["t"] = 0
        del robosim.LOG[:]
        obs0, _ = env.reset()
        rec = dict(reset_state=sim.state.copy(), obs0=np.array(obs0), actions=[], cmds=[], states=[], obs=[],
                   reward=[], done=[], info=[])
        for t in range(T):
            a = rng.uniform(-1, 1, 2).astype(np.float32)
            o, r, d, tr, info = env.step(a)
            rec["actions"].append(a); rec["cmds"].append(robosim.LOG[-1][1]); rec["states"].append(sim.state.copy())
            rec["obs"].append(o); rec["reward"].append(r); rec["done"].append(d)
            rec["info"].append([info[k] for k in ("goal_score", "move", "ball_grad", "energy", "goals_blue", "goals_yellow")])
        episodes.append(rec)
    for i, rec in enumerate(episodes):
        for k, v in rec.items():
            out[f"vss_ep{i}_{k}"] = np.array(v)
    out["vss_n_episodes"] = np.array(len(episodes))

    # ------------------------------------------------------------ RSim adapters / frames
    sim.on_reset = sim.on_step = None
    del robosim.LOG[:]
    cmds_in = [Robot(yellow=False, id=2, v_wheel0=1.5, v_wheel1=-2.5), Robot(yellow=True, id=0, v_wheel0=3.0, v_wheel1=4.0),
               Robot(yellow=True, id=2, v_wheel0=-7.0, v_wheel1=0.25)]
    env.rsim.send_commands(cmds_in)
    out["rsim_vss_cmds"] = robosim.LOG[-1][1]
    s = rng.normal(size=41)
    fr = FrameVSS(); fr.parse(s, 3, 3)
    out["frame_vss_state"] = s
    out["frame_vss_fields"] = np.array([[fr.ball.x, fr.ball.y, fr.ball.z, fr.ball.v_x, fr.ball.v_y, 0]] +
                                       [[r.x, r.y, r.theta, r.v_x, r.v_y, r.v_theta] for r in
                                        [fr.robots_blue[i] for i in range(3)] + [fr.robots_yellow[i] for i in range(3)]])
    random.seed(3)
    frp = env._get_initial_positions_frame()
    frp.ball.v_x, frp.ball.v_y = 0.3, -0.4
    del robosim.LOG[:]
    env.rsim.reset(frp)
    out["rsim_reset_ball"], out["rsim_reset_blue"], out["rsim_reset_yellow"] = robosim.LOG[-1][1:]
    out["rsim_reset_frame"] = np.concatenate([a.ravel() for a in frame_to_arrays(frp, 3, 3)])

    # ------------------------------------------------------------ SSLStaticDefenders-v0
    senv = SSLHWStaticDefendersEnv(field_type=2)
    ssim = senv.rsim.simulator
    out["ssl_field"] = np.array([ssl_field[k] for k in FIELD_KEYS])
    out["sd_norms"] = np.array([senv.max_pos, senv.max_v, senv.max_w])
    out["sd_scales"] = np.array([senv.ball_dist_scale, senv.ball_grad_scale, senv.energy_scale])
    K = 40
    sd_states = np.zeros((K, 82))
    sd_states[:, 0:2] = rng.uniform(-3.5, 3.5, (K, 2)); sd_states[:, 2] = 0.0215
    sd_states[:, 3:5] = rng.uniform(-4, 4, (K, 2))
    for k in range(7):
        o = 5 + 11 * k
        sd_states[:, o:o + 2] = rng.uniform(-3.5, 3.5, (K, 2))
        sd_states[:, o + 2] = rng.uniform(-200, 200, K)
        sd_states[:, o + 3:o + 5] = rng.uniform(-3, 3, (K, 2))
        sd_states[:, o + 5] = rng.uniform(-600, 600, K)
        sd_states[:, o + 6] = rng.integers(0, 2, K)
        sd_states[:, o + 7:o + 11] = rng.uniform(-160, 160, (K, 4))
    obs = []
    for s in sd_states:
        ssim.state = s
        senv.frame = senv.rsim.get_frame()
        obs.append(senv._frame_to_observations())
    out["sd_obs_states"] = sd_states
    out["sd_obs"] = np.array(obs, dtype=np.float32)
    s = rng.normal(size=82); s[5 + 6::11] = rng.integers(0, 2, 7)
    fr = FrameSSL(); fr.parse(s, 1, 6)
    out["frame_ssl_state"] = s
    rl = [fr.robots_blue[0]] + [fr.robots_yellow[i] for i in range(6)]
    out["frame_ssl_fields"] = np.array([[r.x, r.y, r.theta, r.v_x, r.v_y, r.v_theta, float(r.infrared), r.v_wheel0, r.v_wheel1, r.v_wheel2, r.v_wheel3] for r in rl])
    # action -> command (global->local rotation, norm clip, kick, dribbler) through send_commands
    sd_acts = np.concatenate([rng.uniform(-1, 1, (30, 5)), np.array([[1, 1, 1, 1, 1], [-1, 1, 0, -1, -1], [0.6, 0.8, 0.2, 0.0, 0.0]])])
    thetas = rng.uniform(-180, 180, len(sd_acts))
    cm = []
    for a, th in zip(sd_acts, thetas):
        st = sd_states[0].copy(); st[5 + 2] = th
        ssim.state = st
        senv.frame = senv.rsim.get_frame()
        del robosim.LOG[:]
        senv.rsim.send_commands(senv._get_commands(a))
        cm.append(robosim.LOG[-1][1])
    out["sd_cmd_actions"] = sd_acts; out["sd_cmd_thetas"] = thetas; out["sd_cmds"] = np.array(cm)
    # wheel-speed command packing of the adapter
    del robosim.LOG[:]
    senv.rsim.send_commands([Robot(yellow=True, id=3, wheel_speed=True, v_wheel0=1, v_wheel1=2, v_wheel2=3, v_wheel3=4,
                                   kick_v_x=5, kick_v_z=6, dribbler=True)])
    out["rsim_ssl_wheel_cmds"] = robosim.LOG[-1][1]
    # seeded placement
    pl = []
    for sd in seeds:
        random.seed(sd)
        fr = senv._get_initial_positions_frame()
        b, bl, ye = frame_to_arrays(fr, 1, 6)
        pl.append(np.concatenate([b, bl.ravel(), ye.ravel()]))
    out["sd_place"] = np.array(pl)
    # episodes
    sorc = O.OracleEnv(1, 2, 1, 6, 25, "f64")
    sinject = {}

    def s_on_reset(ball, blue, yellow):
        sorc.reset(ball, blue, yellow)
        ssim.state = sorc.get_state()

    def s_on_step(cmds):
        sorc.step(cmds)
        st = sorc.get_state()
        if tick["t"] in sinject:
            st = st.copy()
            for idx, val in sinject[tick["t"]].items():
                st[idx] = val
            full = sorc.get_state_full(); full[:82] = st; sorc.set_state_full(full)
        ssim.state = st
        tick["t"] += 1

    ssim.on_reset, ssim.on_step = s_on_reset, s_on_step
    sd_eps = []
    scripts = [(30, {}), (12, {11: {5: -0.25}}), (12, {11: {5: 2.5, 6: 0.3}}), (12, {11: {0: -0.1}}),
               (12, {11: {0: 3.05, 1: 0.2}}), (12, {11: {0: 3.05, 1: 0.9}}), (12, {11: {1: 2.1}})]
    info_keys = ("goal", "rbt_in_gk_area", "done_ball_out", "done_ball_out_right", "done_rbt_out", "ball_dist", "ball_grad", "energy")
    for ep, (T, inj) in enumerate(scripts):
        random.seed(300 + ep)
        sinject.clear(); sinject.update(inj); tick["t"] = 0
        del robosim.LOG[:]
        obs0, _ = senv.reset()
        rec = dict(reset_state=ssim.state.copy(), obs0=np.array(obs0), actions=[], cmds=[], states=[], obs=[], reward=[], done=[], info=[])
        for t in range(T):
            a = rng.uniform(-1, 1, 5).astype(np.float32)
            if ep == 0:
                a[0] = abs(a[0])  # drive towards the ball side
            o, r, d, tr, info = senv.step(a)
            rec["actions"].append(a); rec["cmds"].append(robosim.LOG[-1][1]); rec["states"].append(ssim.state.copy())
            rec["obs"].append(o); rec["reward"].append(r); rec["done"].append(d)
            rec["info"].append([info[k] for k in info_keys])
            if d:
                break
        sd_eps.append(rec)
    for i, rec in enumerate(sd_eps):
        for k, v in rec.items():
            out[f"sd_ep{i}_{k}"] = np.array(v)
    out["sd_n_episodes"] = np.array(len(sd_eps))

    # ------------------------------------------------------------ the other three SSL tasks
    from rsoccer_gym.ssl.ssl_hw_challenge.dribbling import SSLHWDribblingEnv
    from rsoccer_gym.ssl.ssl_hw_challenge.contested_possession import SSLContestedPossessionEnv
    from rsoccer_gym.ssl.ssl_hw_challenge.pass_endurance import SSLPassEnduranceEnv

    def run_task(tag, env_cls, nb, ny, scripts, info_keys, act_dim, action_hook=None):
        tenv = env_cls()
        tsim = tenv.rsim.simulator
        N = nb + ny
        width = 5 + 11 * N
        torc = O.OracleEnv(1, 2, nb, ny, 25, "f64")
        tinj = {}

        def t_on_reset(ball, blue, yellow):
            torc.reset(ball, np.asarray(blue, float).reshape(-1), np.asarray(yellow, float).reshape(-1))
            tsim.state = torc.get_state()

        def t_on_step(cmds):
            torc.step(cmds)
            st = torc.get_state()
            if tick["t"] in tinj:
                st = st.copy()
                for idx, val in tinj[tick["t"]].items():
                    st[idx] = val
                full = torc.get_state_full(); full[:width] = st; torc.set_state_full(full)
            tsim.state = st
            tick["t"] += 1

        tsim.on_reset, tsim.on_step = t_on_reset, t_on_step
        out[f"{tag}_norms"] = np.array([tenv.max_pos, tenv.max_v, tenv.max_w])
        # observations of arbitrary states
        K = 24
        sts = np.zeros((K, width))
        sts[:, 0:2] = rng.uniform(-3.5, 3.5, (K, 2)); sts[:, 2] = 0.0215; sts[:, 3:5] = rng.uniform(-4, 4, (K, 2))
        for k in range(N):
            o = 5 + 11 * k
            sts[:, o:o + 2] = rng.uniform(-3.5, 3.5, (K, 2)); sts[:, o + 2] = rng.uniform(-200, 200, K)
            sts[:, o + 3:o + 5] = rng.uniform(-3, 3, (K, 2)); sts[:, o + 5] = rng.uniform(-600, 600, K)
            sts[:, o + 6] = rng.integers(0, 2, K); sts[:, o + 7:o + 11] = rng.uniform(-160, 160, (K, 4))
        ob = []
        for i, s_ in enumerate(sts):
            tsim.state = s_
            tenv.frame = tenv.rsim.get_frame()
            if hasattr(tenv, "checkpoints_count"):
                tenv.checkpoints_count = i % 7
            ob.append(tenv._frame_to_observations())
        out[f"{tag}_obs_states"] = sts; out[f"{tag}_obs"] = np.array(ob, dtype=np.float32)
        # seeded placement
        pl = []
        for sd in seeds:
            random.seed(sd)
            fr = tenv._get_initial_positions_frame()
            pl.append(np.concatenate([a.ravel() for a in frame_to_arrays(fr, nb, ny)]))
        out[f"{tag}_place"] = np.array(pl)
        # episodes
        for ep, (T, inj) in enumerate(scripts):
            random.seed(400 + ep)
            tinj.clear(); tinj.update(inj); tick["t"] = 0
            del robosim.LOG[:]
            obs0, _ = tenv.reset()
            rec = dict(reset_state=tsim.state.copy(), obs0=np.array(obs0), actions=[], cmds=[], states=[], obs=[], reward=[], done=[], info=[])
            for t in range(T):
                a = rng.uniform(-1, 1, act_dim).astype(np.float32)
                if action_hook:
                    a = action_hook(ep, t, a)
                rec["actions"].append(a.copy())
                o, r, d, tr, info = tenv.step(a)
                rec["cmds"].append(robosim.LOG[-1][1]); rec["states"].append(tsim.state.copy())
                rec["obs"].append(o); rec["reward"].append(r); rec["done"].append(d)
                rec["info"].append([info[k] for k in info_keys] if info_keys else [0.0])
                if d:
                    break
            for k, v in rec.items():
                out[f"{tag}_ep{ep}_{k}"] = np.array(v)
        out[f"{tag}_n_episodes"] = np.array(len(scripts))

    # dribbling: ball index 0,1; blue0 at 5..; yellow k at 16+11k.  Scripts walk the ball through
    # the checkpoints by teleporting it (the reward code only looks at ball / robot positions).
    def zigzag(points, start=2):
        return {start + i: {0: x, 1: y} for i, (x, y) in enumerate(points)}
    drib_scripts = [
        (40, {}),
        (20, zigzag([(-0.75, 0.2), (-0.75, -0.2), (-1.25, -0.2), (-1.25, 0.2), (-1.75, 0.2), (-1.75, -0.2),
                     (-2.5, -0.2), (-2.5, 0.2), (-1.75, 0.2), (-1.75, -0.2), (-2.5, -0.2), (-2.5, 0.2),
                     (-1.75, 0.2), (-1.75, -0.2)])),
        (12, zigzag([(-0.75, 0.2), (-0.75, -0.2), (-1.25, -0.2), (-1.25, 0.2), (-1.25, -0.2), (-1.75, -0.2), (-1.75, 0.2)])),  # reversed
        (8, {5: {5: 1.5}}),                   # robot leaves the course
        (8, {5: {16 + 3: 0.2}}),              # an obstacle was hit (yellow 0 v_x)
    ]
    def drib_act(ep, t, a):
        a[3] = 1.0
        return a
    run_task("drib", SSLHWDribblingEnv, 1, 4, drib_scripts, None, 4, drib_act)

    cont_keys = ("goal", "rbt_in_gk_area", "done_ball_out", "done_ball_out_right", "done_rbt_out",
                 "ball_dist", "ball_grad", "energy", "collision")
    cont_scripts = [(30, {}), (10, {9: {16 + 3: 0.3}}), (10, {9: {5: -0.3}}), (10, {9: {5: 2.5, 6: 0.1}}),
                    (10, {9: {0: -0.05}}), (10, {9: {0: 3.1, 1: 0.1}}), (10, {9: {0: 3.1, 1: 1.2}}),
                    (10, {9: {16 + 4: -0.2, 0: 3.1, 1: 0.0}})]
    run_task("cont", SSLContestedPossessionEnv, 1, 1, cont_scripts, cont_keys, 5)

    def pass_act(ep, t, a):
        if ep == 0:
            a[0] = 0.3; a[1] = 1.0 if t > 8 else 0.0; a[2] = 1.0
        return a
    pass_scripts = [(60, {}), (40, {}), (30, {10: {0: 3.4, 1: 1.9}}), (12, {11: {16 + 6: 1.0}})]
    run_task("pass", SSLPassEnduranceEnv, 2, 0, pass_scripts, ("reversed_dist", "ball_grad"), 3, pass_act)

    # ------------------------------------------------------------ KD-tree (Utils/kdtree.py)
    pts = rng.uniform(-1, 1, (12, 2))
    qs = rng.uniform(-1, 1, (20, 2))
    tree = KDTree()
    for p in pts:
        tree.insert(tuple(p))
    near = [tree.get_nearest(tuple(q)) for q in qs]
    out["kd_points"] = pts; out["kd_queries"] = qs
    out["kd_nearest"] = np.array([list(n[0]) + [n[1]] for n in near])

    # ------------------------------------------------------------ registry
    reg = gymnasium.envs.registration.REGISTRY
    importlib.import_module("rsoccer_gym")
    with open(os.path.join(HERE, "registry.json"), "w") as f:
        json.dump({k: dict(max_episode_steps=v["max_episode_steps"], kwargs=v["kwargs"], entry_point=v["entry_point"]) for k, v in reg.items()}, f, indent=1, sort_keys=True)

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", len(out), "arrays ->", os.path.join(HERE, "reference_vectors.npz"),
          os.path.getsize(os.path.join(HERE, "reference_vectors.npz")), "bytes")


if __name__ == "__main__":
    main()
