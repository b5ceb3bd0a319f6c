"""GPU <-> CPU-oracle parity through the C-ABI (librsx_hip.so).  Bit-exact in fp32.

The oracle (oracle/rsx_oracle.c, float instantiation) and the HIP kernels are two independent
implementations of the step model; every comparison below is exact equality of float32 bit
patterns, over whole trajectories (collisions, goals, auto-resets, RNG included).
"""
import os

import numpy as np
import pytest

from helpers import f32_equal, mismatch_report, random_placement

pytestmark = pytest.mark.gpu


def _lib():
    from rsoccer_amd import _lib
    return _lib


def _mk_oracles(O, kind, ft, nb, ny, B, ts=25):
    return [O.OracleEnv(kind, ft, nb, ny, ts, "f32") for _ in range(B)]


CASES = [
    # kind, field_type, nb, ny, B, steps, spread
    (0, 0, 3, 3, 67, 60, 0.5),     # VSS 3v3 (ragged batch: 67 is not a multiple of 8)
    (0, 1, 5, 5, 33, 40, 0.5),     # VSS 5v5 -> 16 lanes per env
    (1, 2, 1, 6, 41, 60, 0.4),     # SSL 1v6
    (1, 0, 6, 6, 21, 40, 0.3),     # SSL 6v6 -> 16 lanes per env, fixed-size variant
    (1, 0, 4, 3, 11, 30, 0.3),     # SSL 4v3 -> run-time robot count
    (1, 1, 11, 11, 19, 40, 0.25),  # SSL 11v11 -> 32 lanes per env, crowded
    (1, 0, 2, 0, 5, 30, 0.2),      # SSL 2v0 (empty yellow team)
]


@pytest.mark.parametrize("kind,ft,nb,ny,B,steps,spread", CASES)
def test_raw_step_bitexact(oracle_mod, kind, ft, nb, ny, B, steps, spread):
    L = _lib()
    O = oracle_mod
    rng = np.random.default_rng(1234 + kind * 10 + nb)
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    fp = sim.get_field_params()
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    assert np.allclose(list(fp.values()), refs[0].field_params())
    N = nb + ny
    # dummy line-up right after construction
    st = sim.get_state()
    for e in range(B):
        assert f32_equal(st[e], refs[e].get_state())
    min_d = 2.2 * fp["rbt_radius"]
    ball, blue, yellow = random_placement(rng, B, nb, ny, fp["length"] / 2, fp["width"] / 2, min_d, spread)
    sim.reset(ball, blue, yellow if ny else None)
    for e in range(B):
        refs[e].reset(ball[e], blue[e], yellow[e] if ny else np.zeros(0))
    C = sim.cmd_dim
    contacts = 0
    for t in range(steps):
        if kind == 0:
            cmds = rng.uniform(-60, 60, (B, N, 2))
        else:
            cmds = np.zeros((B, N, 8))
            use_wheels = rng.random((B, N)) < 0.3
            cmds[..., 0] = use_wheels
            cmds[..., 1:5] = np.where(use_wheels[..., None], rng.uniform(-120, 120, (B, N, 4)),
                                      np.concatenate([rng.uniform(-3, 3, (B, N, 2)),
                                                      rng.uniform(-12, 12, (B, N, 1)),
                                                      np.zeros((B, N, 1))], -1))
            cmds[..., 5] = np.where(rng.random((B, N)) < 0.3, 4.0, 0.0)
            cmds[..., 6] = np.where(rng.random((B, N)) < 0.1, 2.0, 0.0)
            cmds[..., 7] = rng.random((B, N)) < 0.5
        sim.step(cmds)
        got = sim.get_state_full()
        for e in range(B):
            refs[e].step(cmds[e])
            want = refs[e].get_state_full()
            assert f32_equal(got[e], want), mismatch_report(got[e], want, f"env {e} step {t}")
    sim.close()


@pytest.mark.parametrize("kind,ft,nb,ny,B", [(0, 0, 3, 3, 70), (1, 1, 11, 11, 21), (1, 2, 1, 6, 33)])
def test_raw_step_with_device_random_commands_bitexact(oracle_mod, kind, ft, nb, ny, B):
    """rsx_step_dev_random: commands drawn in the kernel (Philox, SURVEY.md 8(d) config 4 distribution)
    equal the oracle's restatement; the tick range continues across calls."""
    L = _lib()
    rng = np.random.default_rng(12)
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    f = sim.get_field_params()
    ball, blue, yellow = random_placement(rng, B, nb, ny, f["length"] / 2 - 0.3, f["width"] / 2 - 0.3, 2.3 * f["rbt_radius"], 0.5)
    sim.reset(ball, blue, yellow)
    refs = _mk_oracles(oracle_mod, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.reset(ball[e], blue[e], yellow[e])
    seed = 0xABCDEF0123
    sim.step_dev_random(25, seed, 0)
    sim.step_dev_random(15, seed, 25)
    for e, r in enumerate(refs):
        for t in range(40):
            r.step_random(seed, e, t)
    got = sim.get_state_full()
    for e, r in enumerate(refs):
        assert f32_equal(got[e], r.get_state_full()), mismatch_report(got[e], r.get_state_full(), f"env {e}")
    moved = np.abs(got[:, 5] - np.array([b[0, 0] if nb else 0 for b in blue])).max()
    assert moved > 0.05
    sim.close()


def test_set_get_state_roundtrip():
    L = _lib()
    sim = L.Sim(1, 2, 1, 6, 25, 9)
    rng = np.random.default_rng(0)
    s = rng.normal(size=(9, sim.state_dim + 2)).astype(np.float32).astype(np.float64)
    sim.set_state(s)
    assert np.array_equal(sim.get_state_full(), s)
    assert np.array_equal(sim.get_state(), s[:, :-2])
    sim.close()


def _cmp_task(sim, refs, tens, t, check_state=True):
    import torch
    torch.cuda.synchronize()
    obs = tens["obs"].cpu().numpy()
    rew = tens["reward"].cpu().numpy()
    term = tens["terminated"].cpu().numpy()
    trunc = tens["truncated"].cpu().numpy()
    info = tens["info"].cpu().numpy()
    steps = tens["steps"].cpu().numpy()
    fin = tens["final_obs"].cpu().numpy()
    state = sim.get_state_full() if check_state else None
    for e, r in enumerate(refs):
        o = r.task_out()
        assert f32_equal(obs[e], o["obs"]), mismatch_report(obs[e], o["obs"], f"obs env {e} step {t}")
        assert f32_equal(rew[e], o["reward"]), f"reward env {e} step {t}: {rew[e]} vs {o['reward']}"
        assert term[e] == o["terminated"] and trunc[e] == o["truncated"], f"done env {e} step {t}"
        assert f32_equal(info[:, e], o["info"]), mismatch_report(info[:, e], o["info"], f"info env {e} step {t}")
        assert steps[e] == o["steps"]
        if o["terminated"] or o["truncated"]:
            assert f32_equal(fin[e], o["final_obs"]), f"final_obs env {e} step {t}"
        if check_state:
            w = r.get_state_full()
            assert f32_equal(state[e], w), mismatch_report(state[e], w, f"state env {e} step {t}")


TASKS = [
    # task, kind, ft, nb, ny, B, steps, max_episode_steps
    (1, 0, 0, 3, 3, 45, 260, 100),     # VSS-v0
    (2, 1, 2, 1, 6, 37, 200, 60),      # SSLStaticDefenders-v0
    (3, 1, 2, 1, 4, 29, 220, 70),      # SSLDribbling-v0
    (4, 1, 2, 1, 1, 53, 220, 50),      # SSLContestedPossession-v0
    (5, 1, 2, 2, 0, 21, 220, 40),      # SSLPassEndurance-v0
    (6, 1, 1, 11, 11, 13, 90, 35),     # scrimmage 11v11, division-A field, spread line-up (32 lanes per env, 22 robots unrolled)
    (7, 1, 1, 11, 11, 11, 90, 35),     # scrimmage 11v11, crowded line-up: contacts in every sub-step
    (7, 1, 2, 3, 2, 27, 120, 30),      # scrimmage 3v2 on the small field (8 lanes per env, run-time robot count)
    (6, 1, 0, 6, 6, 9, 80, 40),        # scrimmage 6v6 (16 lanes per env)
]


@pytest.mark.parametrize("task,kind,ft,nb,ny,B,steps,max_steps", TASKS)
def test_task_random_actions_bitexact(oracle_mod, task, kind, ft, nb, ny, B, steps, max_steps):
    """Whole fused rollouts with device-side random actions, OU noise, TimeLimit and auto-reset."""
    L = _lib()
    O = oracle_mod
    seed, base = 0x1234567890ABCDEF, 1000
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(task, seed, base, max_steps)
    tens = sim.task_tensors()
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(task, seed, base + e, max_steps)
        r.task_reset()
    sim.task_reset()
    import torch
    torch.cuda.synchronize()
    obs = tens["obs"].cpu().numpy()
    st = sim.get_state_full()
    for e, r in enumerate(refs):
        assert f32_equal(st[e], r.get_state_full()), mismatch_report(st[e], r.get_state_full(), f"reset state env {e}")
        assert f32_equal(obs[e], r.task_out()["obs"])
    for t in range(steps):
        sim.task_step(None)
        for r in refs:
            r.task_step(None)
        _cmp_task(sim, refs, tens, t)
    got = sim.read_metrics()
    want = sum(r.task_out()["metrics"] for r in refs)
    assert np.array_equal(got, want), (got, want)
    assert got[1] > 0  # episodes did end (TimeLimit at least)
    sim.close()


@pytest.mark.parametrize("task,kind,ft,nb,ny,B,steps,max_steps", TASKS)
def test_task_fed_actions_and_modes_bitexact(oracle_mod, task, kind, ft, nb, ny, B, steps, max_steps):
    """Host-chosen actions; step_n (n launches from C) and rollout (one launch) agree with single steps."""
    import torch
    L = _lib()
    O = oracle_mod
    seed, base = 7, 0
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(task, seed, base, max_steps)
    tens = sim.task_tensors()
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(task, seed, base + e, max_steps)
        r.task_reset()
    sim.task_reset()
    rng = np.random.default_rng(5)
    A = sim.act_dim
    for t in range(40):
        a = rng.uniform(-1, 1, (B, A)).astype(np.float32)
        tens["actions"].copy_(torch.from_numpy(a))
        sim.task_step(tens["actions"].data_ptr())
        for e, r in enumerate(refs):
            r.task_step(a[e])
        _cmp_task(sim, refs, tens, t)
    sim.task_step_n(17)
    for r in refs:
        for _ in range(17):
            r.task_step(None)
    _cmp_task(sim, refs, tens, "graph")
    sim.task_rollout(23)
    for r in refs:
        for _ in range(23):
            r.task_step(None)
    _cmp_task(sim, refs, tens, "rollout")
    sim.close()


@pytest.mark.parametrize("task,kind,ft,nb,ny", [(1, 0, 0, 3, 3), (2, 1, 2, 1, 6), (1, 0, 1, 5, 5)])
def test_many_resets_placement_bitexact(oracle_mod, task, kind, ft, nb, ny):
    """TimeLimit = 1: every env is re-placed in every step, 20k+ placements per case.  The kernel
    places in parallel rounds (one per rejection), the oracle with the reference's sequential
    rejection loop (vss_gym.py:194-233, static_defenders.py:214-254); this many placements reach
    several rejections in one env and draw indices beyond the pre-drawn block."""
    import torch
    L = _lib()
    O = oracle_mod
    B, steps, seed = 2048, 10, 99
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(task, seed, 0, 1)
    tens = sim.task_tensors()
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(task, seed, e, 1)
        r.task_reset()
    sim.task_reset()
    for t in range(steps):
        sim.task_step(None)
        O.vec_task_step(refs, 1)
        torch.cuda.synchronize()
        st = sim.get_state_full()
        obs = tens["obs"].cpu().numpy()
        want = np.stack([r.get_state_full() for r in refs])
        assert f32_equal(st, want), mismatch_report(st, want, f"state after step {t}")
        wobs = np.stack([r.task_out()["obs"] for r in refs])
        assert f32_equal(obs, wobs), mismatch_report(obs, wobs, f"obs after step {t}")
    assert sim.read_metrics()[1] == B * steps
    sim.close()


@pytest.mark.parametrize("task,kind,ft,nb,ny,B,steps", [(1, 0, 0, 3, 3, 4096, 6000), (2, 1, 2, 1, 6, 2048, 3000),
                                                       (3, 1, 2, 1, 4, 2048, 2000), (4, 1, 2, 1, 1, 2048, 2000),
                                                       (5, 1, 2, 2, 0, 2048, 2000),
                                                       (7, 1, 1, 11, 11, 1024, 1500)])   # configs[3], crowded, full size
def test_full_size_soak_invariants(task, kind, ft, nb, ny, B, steps):
    """BASELINE.json's batch sizes, thousands of steps (tens of millions of env-steps, thousands of
    contacts and resets): size-independent properties instead of the oracle.  Everything stays
    finite, bodies stay inside the walls, observations inside the clip range, the counters add up,
    and a second run from the same seed reproduces every buffer bit for bit."""
    import torch
    L = _lib()
    out = []
    for rep in range(2):
        sim = L.Sim(kind, ft, nb, ny, 25, B)
        sim.task_attach(task, 4242, 0, 0)
        tens = sim.task_tensors()
        sim.task_reset()
        done_seen = 0
        for chunk in range(steps // 500):
            sim.task_step_n(499)
            sim.task_step(None)
            torch.cuda.synchronize()
            done_seen += int((tens["terminated"] | tens["truncated"]).sum().item())   # sampled every 500th step
            obs = tens["obs"]
            assert torch.isfinite(obs).all() and obs.abs().max().item() <= float(np.float32(1.2))
            assert torch.isfinite(tens["reward"]).all()
        st = sim.get_state_full()
        assert np.isfinite(st).all()
        f = sim.get_field_params()
        margin = 0.0 if kind == 0 else 0.35
        xmax = f["length"] / 2 + f["goal_depth"] + margin + 1e-4
        ymax = f["width"] / 2 + margin + 1e-4
        N = nb + ny
        rs = 6 if kind == 0 else 11
        xs = np.concatenate([st[:, 0:1]] + [st[:, 5 + rs * k: 6 + rs * k] for k in range(N)], 1)
        ys = np.concatenate([st[:, 1:2]] + [st[:, 6 + rs * k: 7 + rs * k] for k in range(N)], 1)
        assert np.abs(xs).max() <= xmax and np.abs(ys).max() <= ymax, (np.abs(xs).max(), np.abs(ys).max())
        th = np.stack([st[:, 7 + rs * k] for k in range(N)], 1)
        assert np.abs(th).max() <= 360.0
        m = sim.read_metrics()
        assert m[0] == B * steps and m[1] > 0 and m[5] <= m[0] and m[6] <= m[1]
        out.append(np.concatenate([st.ravel(), tens["obs"].cpu().numpy().astype(np.float64).ravel(), m.astype(np.float64)]))
        sim.close()
    assert np.array_equal(out[0], out[1])


RR_WORST_CM, RR_P99_CM = 1.5, 0.7   # model v2; measured on MI355X: 1.32 / 0.47 (model v1: 4.0-4.5 / 1.7-2.0)


def test_crowded_11v11_full_size_contact_invariants():
    """BASELINE.json configs[3] at its full size: 1024 envs x 22 robots on the division-A field, all of
    them chasing the ball (a 22-robot scrum: the worst case for the all-pairs contact sweeps), random
    kicks and dribblers, 1000 steps.  Size-independent properties: everything finite and inside the
    walls, robot-robot overlap below a tenth of a diameter even in a pile that the robots' own push presses against a
    wall, into a corner or into a goal (model v2: wall-aware shares, third and fourth sweep at walls, goal posts — DESIGN.md 4,
    profiles/r05_jam_model_v2.txt; the CPU definition of the model gives worst 1.32 cm, p99 0.47 cm over 256 envs x 1000 steps),
    and the ball centre never deep inside a kicker face."""
    import torch
    L = _lib()
    B, N = 1024, 22
    sim = L.Sim(1, 1, 11, 11, 25, B)
    rng = np.random.default_rng(3)
    grid = np.array([(0.2 * (i - 2.5), 0.2 * (j - 1.5)) for i in range(6) for j in range(4)][:N])
    pose = np.zeros((B, N, 3))
    pose[:, :, :2] = grid[None] + rng.uniform(-0.008, 0.008, (B, N, 2))
    pose[:, :, 2] = rng.uniform(-180, 180, (B, N))
    ball = np.zeros((B, 4)); ball[:, 1] = 0.1
    sim.reset(ball, pose[:, :11], pose[:, 11:])
    st = sim.state_tensor()
    cm = sim.cmds_tensor().view(N, 8, B)
    rows = torch.arange(N, device="cuda") * 11 + 5
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    f = sim.get_field_params()
    worst_rr = worst_rb = 0.0
    samples = []   # deepest robot-robot overlap of every env, every fifth step
    eye = torch.eye(N, device="cuda", dtype=torch.bool)[:, :, None]
    for t in range(1000):
        x, y, th = st[rows], st[rows + 1], torch.deg2rad(st[rows + 2])
        gx, gy = st[0][None] - x, st[1][None] - y
        n = torch.hypot(gx, gy) + 1e-9
        gx, gy = 2.0 * gx / n, 2.0 * gy / n
        cm.zero_()
        cm[:, 1] = gx * torch.cos(th) + gy * torch.sin(th)
        cm[:, 2] = -gx * torch.sin(th) + gy * torch.cos(th)
        cm[:, 3] = torch.rand(N, B, device="cuda", generator=gen) * 6 - 3
        cm[:, 5] = (torch.rand(N, B, device="cuda", generator=gen) > 0.9) * 3.0
        cm[:, 7] = (torch.rand(N, B, device="cuda", generator=gen) > 0.5).float()
        sim.step_dev()
        if t % 5 == 4:
            x, y = st[rows], st[rows + 1]
            d = torch.hypot(x[:, None] - x[None], y[:, None] - y[None]).masked_fill(eye, 9.0)
            samples.append(0.18 - d.amin(dim=(0, 1)))
            low = st[2] < 0.15
            db = torch.hypot(x - st[0][None], y - st[1][None]).min(0).values
            worst_rb = max(worst_rb, ((0.073 + 0.0215) - db)[low].max().item())
    torch.cuda.synchronize()
    full = sim.get_state_full()
    assert np.isfinite(full).all()
    xs = np.concatenate([full[:, 0:1]] + [full[:, 5 + 11 * k: 6 + 11 * k] for k in range(N)], 1)
    ys = np.concatenate([full[:, 1:2]] + [full[:, 6 + 11 * k: 7 + 11 * k] for k in range(N)], 1)
    assert np.abs(xs).max() <= f["length"] / 2 + f["goal_depth"] + 0.35 and np.abs(ys).max() <= f["width"] / 2 + 0.35
    over = torch.stack(samples).clamp_min(0.0).flatten()
    worst_rr = over.max().item()
    p99 = torch.quantile(over.float(), 0.99).item()
    if os.environ.get("RSX_PRINT_ENVELOPE"):
        print(f"crowded 11v11 envelope: worst robot-robot overlap {worst_rr * 100:.2f} cm, p99 {p99 * 100:.2f} cm, "
              f"worst ball-in-kicker {worst_rb * 100:.2f} cm")
    # In the open the contact model holds 4-5 mm; what is left above that sits at walls, corners and goals (tools/exp_jam2.py).  A change
    # that makes the jam behaviour worse has to fail here.
    msg = (f"worst robot-robot overlap {worst_rr * 100:.2f} cm (bound %s), p99 {p99 * 100:.2f} cm (bound %s): the wall-pile "
           "residual of the contact model grew — see DESIGN.md 4") % (RR_WORST_CM, RR_P99_CM)
    assert 0.005 < worst_rr < RR_WORST_CM / 100.0, msg      # it IS a scrum, and nothing tunnels
    assert p99 < RR_P99_CM / 100.0, msg
    assert worst_rb < 0.04, worst_rb
    sim.close()


def _vss_scrum(L, ft, nb, B, steps, seed):
    """every robot of a VSS field drives at the ball (proportional heading control on the wheel commands, a little steering noise);
    returns the deepest robot-robot overlap of every env after every step and the final full state"""
    import torch
    N = 2 * nb
    sim = L.Sim(0, ft, nb, nb, 25, B)
    f = sim.get_field_params()
    rng = np.random.default_rng(seed)
    cols = (N + 1) // 2
    pose = np.zeros((B, N, 3))
    k = np.arange(N)
    pose[:, :, 0] = ((k % cols) - (cols - 1) / 2.0) * (f["length"] * 0.7 / cols) + rng.uniform(-0.01, 0.01, (B, N))
    pose[:, :, 1] = np.where(k // cols > 0, 0.3, -0.3) + rng.uniform(-0.01, 0.01, (B, N))
    pose[:, :, 2] = rng.uniform(-180, 180, (B, N))
    ball = np.zeros((B, 4))
    ball[:, 0] = rng.uniform(-0.4, 0.4, B) * f["length"]; ball[:, 1] = rng.uniform(-0.15, 0.15, B) * f["width"]
    sim.reset(ball, pose[:, :nb], pose[:, nb:])
    st = sim.state_tensor()
    cm = sim.cmds_tensor().view(N, 2, B)
    rows = torch.arange(N, device="cuda") * 6 + 5
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    eye = torch.eye(N, device="cuda", dtype=torch.bool)[:, :, None]
    samples = []
    for t in range(steps):
        x, y, th = st[rows], st[rows + 1], torch.deg2rad(st[rows + 2])
        err = torch.atan2(st[1][None] - y, st[0][None] - x) - th
        err = torch.atan2(torch.sin(err), torch.cos(err))
        v = 0.9 * torch.cos(err).clamp_min(0.0) + 0.1
        w = 8.0 * err + (torch.rand(N, B, device="cuda", generator=gen) - 0.5) * 4.0
        cm[:, 0] = (v - w * 0.04) / 0.026
        cm[:, 1] = (v + w * 0.04) / 0.026
        sim.step_dev()
        x, y = st[rows], st[rows + 1]
        d = torch.hypot(x[:, None] - x[None], y[:, None] - y[None]).masked_fill(eye, 9.0)
        samples.append(0.075 - d.amin(dim=(0, 1)))
    torch.cuda.synchronize()
    full = sim.get_state_full()
    sim.close()
    return torch.stack(samples).clamp_min(0.0).flatten().float(), full, f


# model v2 for the VSS class (profiles/r06_jam_vss_model_v2.txt).  3v3: the CPU definition gives worst 0.78 cm, p99 0.23 cm over
# 256 envs x 1000 steps (model v1: 3.5 / 0.49); measured on MI355X at 4096 envs: 0.95 / 0.24.  5v5: ten robots pile up in the open as
# well, where the pile sits at the depth that triggers the second sweep (pen2 = 5 mm, like the 22-robot SSL scrum): CPU worst 1.06,
# p99 0.48 (v1: 1.72 / 1.13); MI355X at 1024 envs: 1.45 / 0.47
VSS_ENVELOPE = {(0, 3): (4096, 1.5, 0.3), (1, 5): (1024, 2.0, 0.6)}


@pytest.mark.parametrize("ft,nb", [(0, 3), (1, 5)], ids=["3v3", "5v5"])
def test_vss_scrum_full_size_contact_invariants(ft, nb):
    """BASELINE.json configs[1]'s simulator at its full batch (and the 5v5 field at 1024 envs): every robot drives at the ball for 1000
    steps — on the 1.5 m x 1.3 m field the scrum is at a wall, in a corner or in a goal mouth most of the time.  Size-independent
    properties: everything finite and inside the walls, and the robots stay discs: the deepest robot-robot overlap of every env after
    every step stays inside the envelope of model v2 (held axes + goal posts)."""
    L = _lib()
    B, worst_cm, p99_cm = VSS_ENVELOPE[(ft, nb)]
    over, full, f = _vss_scrum(L, ft, nb, B, 1000, 5)
    N = 2 * nb
    assert np.isfinite(full).all()
    xs = np.concatenate([full[:, 0:1]] + [full[:, 5 + 6 * k: 6 + 6 * k] for k in range(N)], 1)
    ys = np.concatenate([full[:, 1:2]] + [full[:, 6 + 6 * k: 7 + 6 * k] for k in range(N)], 1)
    assert np.abs(xs).max() <= f["length"] / 2 + f["goal_depth"] + 1e-4 and np.abs(ys).max() <= f["width"] / 2 + 1e-4
    import torch
    worst = over.max().item()
    p99 = torch.quantile(over[torch.randperm(over.numel(), device=over.device)[:4_000_000]], 0.99).item()
    touching = (over > 0).float().mean().item()
    if os.environ.get("RSX_PRINT_ENVELOPE"):
        print(f"VSS {nb}v{nb} scrum envelope ({B} envs): touching {100 * touching:.1f} %, worst {worst * 100:.2f} cm, p99 {p99 * 100:.2f} cm")
    msg = f"worst robot-robot overlap {worst * 100:.2f} cm (bound {worst_cm}), p99 {p99 * 100:.2f} cm (bound {p99_cm}): DESIGN.md 4"
    assert touching > 0.8                                   # it IS a scrum
    assert 0.002 < worst < worst_cm / 100.0, msg
    assert p99 < p99_cm / 100.0, msg


def test_batch_position_and_shard_invariance():
    """env i's trajectory depends only on (seed, global env id): not on batch size, position
    in the batch, or how the batch is split over handles (= over GPUs)."""
    import torch
    L = _lib()
    seed = 99
    full = L.Sim(0, 0, 3, 3, 25, 64)
    full.task_attach(1, seed, 0, 50)
    full.task_reset()
    full.task_step_n(120)
    torch.cuda.synchronize()
    want = full.get_state_full()
    wobs = full.task_tensors()["obs"].cpu().numpy()
    parts = []
    for lo, n in ((0, 24), (24, 3), (27, 37)):
        s = L.Sim(0, 0, 3, 3, 25, n)
        s.task_attach(1, seed, lo, 50)
        s.task_reset()
        for _ in range(120):
            s.task_step(None)
        torch.cuda.synchronize()
        parts.append((s.get_state_full(), s.task_tensors()["obs"].cpu().numpy(), s.read_metrics()))
        s.close()
    got = np.concatenate([p[0] for p in parts])
    gobs = np.concatenate([p[1] for p in parts])
    assert np.array_equal(got, want)
    assert np.array_equal(gobs, wobs)
    assert np.array_equal(sum(p[2] for p in parts), full.read_metrics())
    full.close()


def test_reset_to_matches_oracle(oracle_mod):
    import torch
    L = _lib()
    O = oracle_mod
    B = 12
    sim = L.Sim(0, 0, 3, 3, 25, B)
    sim.task_attach(1, 3, 0, 0)
    refs = _mk_oracles(O, 0, 0, 3, 3, B)
    for e, r in enumerate(refs):
        r.task_attach(1, 3, e, 0)
        r.task_reset()
    sim.task_reset()
    rng = np.random.default_rng(2)
    ball, blue, yellow = random_placement(rng, B, 3, 3, 0.6, 0.5, 0.1)
    mask = (rng.random(B) < 0.5).astype(np.uint8)
    sim.task_step_n(5)
    for r in refs:
        for _ in range(5):
            r.task_step(None)
    sim.task_reset_to(ball, blue, yellow, mask)
    for e, r in enumerate(refs):
        if mask[e]:
            r.task_reset_to(ball[e], blue[e], yellow[e])
    tens = sim.task_tensors()
    torch.cuda.synchronize()
    obs = tens["obs"].cpu().numpy()
    for e, r in enumerate(refs):
        if mask[e]:
            assert f32_equal(obs[e], r.task_out()["obs"])
    for t in range(30):
        sim.task_step(None)
        for r in refs:
            r.task_step(None)
    _cmp_task(sim, refs, tens, "after reset_to")
    sim.close()


def test_vss_v0_on_the_5v5_field_uses_16_lane_groups(oracle_mod):
    """VSS-v0 arithmetic on field_type 1 (5v5: 11 bodies -> 16 lanes per env, generic kernel)."""
    L = _lib()
    O = oracle_mod
    B, seed = 19, 21
    sim = L.Sim(0, 1, 5, 5, 25, B)
    sim.task_attach(1, seed, 0, 45)
    assert sim.obs_dim == 64
    tens = sim.task_tensors()
    refs = _mk_oracles(O, 0, 1, 5, 5, B)
    for e, r in enumerate(refs):
        r.task_attach(1, seed, e, 45)
        r.task_reset()
    sim.task_reset()
    for t in range(120):
        sim.task_step(None)
        for r in refs:
            r.task_step(None)
    _cmp_task(sim, refs, tens, "5v5")
    sim.close()


def test_one_wavefront_per_env_layout_gives_identical_results():
    """RSX_LANES_PER_ENV=64 selects the 'one wavefront per env' mapping of BASELINE.json's north
    star; results must not depend on the mapping (run in a subprocess: the variable is read at
    handle creation)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "from rsoccer_amd import _lib as L\n"
        "out = []\n"
        "for task, kind, ft, nb, ny in ((1, 0, 0, 3, 3), (2, 1, 2, 1, 6)):\n"
        "    s = L.Sim(kind, ft, nb, ny, 25, 37); s.task_attach(task, 5, 0, 30); s.task_reset(); s.task_step_n(100)\n"
        "    torch.cuda.synchronize(); out.append(s.get_state_full()); out.append(s.task_tensors()['obs'].cpu().numpy().astype(np.float64))\n"
        "np.save(sys.argv[1], np.concatenate([o.ravel() for o in out]))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = []
    for lanes in (None, "64"):
        env = dict(os.environ)
        env.pop("RSX_LANES_PER_ENV", None)
        if lanes:
            env["RSX_LANES_PER_ENV"] = lanes
        path = "/tmp/rsx_lanes_%s.npy" % (lanes or "default")
        subprocess.check_call([sys.executable, "-c", code, path], env=env)
        res.append(np.load(path))
    assert np.array_equal(res[0], res[1])


def test_get_state_after_device_side_writes_is_not_served_from_the_step_copy():
    """rsx_step brings the new state home for the rsx_get_state that follows it; that copy must
    not survive a change made behind the host API (torch writing through the device view, a
    device-side step, a fused task step)."""
    import torch
    L = _lib()
    sim = L.Sim(0, 0, 3, 3, 25, 5)
    cmds = np.zeros((5, 6, 2))
    sim.step(cmds)
    a = sim.get_state()
    sim.step_dev()                       # device-side step: state changes without the host path
    sim.step(cmds)
    st = sim.state_tensor()              # raw pointers handed out
    st[0, :] = 0.123                     # ball x of every env, written by torch
    torch.cuda.synchronize()
    b = sim.get_state()
    assert np.allclose(b[:, 0], 0.123) and not np.allclose(a[:, 0], 0.123)
    sim.step(cmds)
    st[1, :] = -0.2
    torch.cuda.synchronize()
    assert np.allclose(sim.get_state()[:, 1], -0.2)
    sim.close()


EPL_TASKS = [(1, 0, 0, 3, 3, 2), (2, 1, 2, 1, 6, 5), (3, 1, 2, 1, 4, 4), (4, 1, 2, 1, 1, 5), (5, 1, 2, 2, 0, 3)]
EPL_IDS = ["vss-v0", "static-defenders", "dribbling", "contested", "pass-endurance"]


@pytest.mark.parametrize("lean", [None, "0"], ids=["default-form", "classic-form"])
@pytest.mark.parametrize("task,kind,ft,nb,ny,adim", EPL_TASKS, ids=EPL_IDS)
def test_env_per_lane_layout_is_bit_identical(oracle_mod, monkeypatch, task, kind, ft, nb, ny, adim, lean):
    """Large batches of the five registered tasks are stepped by second kernels (one lane per env,
    rsx_epl.hpp / rsx_epl_ssl.hpp).  Forced on a small ragged batch here: they must agree bit for bit
    with the CPU oracle and with the 8-lanes-per-env kernel — fed and random actions (kicks, dribbler),
    contacts, TimeLimit resets, single-step and multi-step launches, counters."""
    import torch
    L = _lib()
    O = oracle_mod
    B, seed, base, max_steps = 150, 31337, 77, 45
    rng = np.random.default_rng(5)
    outs = {}
    refs = None
    if lean is not None:   # the SSL single-step kernels have two forms of their epilogue (rsx_epl.hip picks by task and batch)
        monkeypatch.setenv("RSX_EPL_LEAN", lean)
    for layout in ("epl", "lanes"):
        monkeypatch.setenv("RSX_LAYOUT", layout)
        sim = L.Sim(kind, ft, nb, ny, 25, B)
        sim.task_attach(task, seed, base, max_steps)
        tens = sim.task_tensors()
        sim.task_reset()
        if layout == "epl":
            refs = _mk_oracles(O, kind, ft, nb, ny, B)
            for e, r in enumerate(refs):
                r.task_attach(task, seed, base + e, max_steps)
                r.task_reset()
        rng = np.random.default_rng(5)
        for t in range(120):
            if t % 3 == 0:
                a = rng.uniform(-1, 1, tuple(tens["actions"].shape)).astype(np.float32)
                assert a[0].size == adim
                tens["actions"].copy_(torch.from_numpy(a))
                sim.task_step(tens["actions"].data_ptr())
                if layout == "epl":
                    for e, r in enumerate(refs):
                        r.task_step(a[e])
            else:
                sim.task_step(None)
                if layout == "epl":
                    for r in refs:
                        r.task_step(None)
            if layout == "epl" and t % 7 == 0:
                _cmp_task(sim, refs, tens, t)
        sim.task_rollout(60)
        if layout == "epl":
            O.vec_task_step(refs, 60)
            _cmp_task(sim, refs, tens, "rollout")
            want = sum(r.task_out()["metrics"] for r in refs)
            assert np.array_equal(sim.read_metrics(), want)
        torch.cuda.synchronize()
        outs[layout] = np.concatenate([sim.get_state_full().ravel()] + [tens[k].cpu().numpy().astype(np.float64).ravel()
                                      for k in ("obs", "reward", "terminated", "truncated", "info", "final_obs", "steps")]
                                      + [sim.read_metrics().astype(np.float64)])
        sim.close()
    assert np.array_equal(outs["epl"], outs["lanes"], equal_nan=True)


@pytest.mark.parametrize("task", [6, 7], ids=["spread", "crowded"])
def test_quad_layout_is_bit_identical(oracle_mod, monkeypatch, task):
    """SSL 11v11 scrimmage (BASELINE.json configs[3]) at large batches is stepped by a kernel with FOUR lanes per env, six
    robots per lane (rsx_quad_ssl.hpp).  Forced on a small ragged batch here: the CPU oracle's bits and the 32-lane kernel's,
    with fed and drawn actions (kicks), contacts in every sub-step (crowded line-up), goals and TimeLimit resets, counters."""
    import torch
    L = _lib()
    O = oracle_mod
    kind, ft, nb, ny = 1, 1, 11, 11
    B, seed, base, max_steps = 75, 4711, 19, 40
    outs = {}
    refs = None
    for layout in ("quad", "lanes"):
        monkeypatch.setenv("RSX_LAYOUT", layout)
        sim = L.Sim(kind, ft, nb, ny, 25, B)
        sim.task_attach(task, seed, base, max_steps)
        tens = sim.task_tensors()
        sim.task_reset()
        if layout == "quad":
            refs = _mk_oracles(O, kind, ft, nb, ny, B)
            for e, r in enumerate(refs):
                r.task_attach(task, seed, base + e, max_steps)
                r.task_reset()
        rng = np.random.default_rng(11)
        for t in range(100):
            if t % 3 == 0:
                a = rng.uniform(-1, 1, tuple(tens["actions"].shape)).astype(np.float32)
                tens["actions"].copy_(torch.from_numpy(a))
                sim.task_step(tens["actions"].data_ptr())
                if layout == "quad":
                    for e, r in enumerate(refs):
                        r.task_step(a[e])
            else:
                sim.task_step(None)
                if layout == "quad":
                    for r in refs:
                        r.task_step(None)
            if layout == "quad" and t % 9 == 0:
                _cmp_task(sim, refs, tens, t)
        if layout == "quad":
            _cmp_task(sim, refs, tens, "end")
            want = sum(r.task_out()["metrics"] for r in refs)
            assert np.array_equal(sim.read_metrics(), want) and want[1] >= B
        torch.cuda.synchronize()
        outs[layout] = np.concatenate([sim.get_state_full().ravel()] + [tens[k].cpu().numpy().astype(np.float64).ravel()
                                      for k in ("obs", "reward", "terminated", "truncated", "info", "final_obs", "steps")]
                                      + [sim.read_metrics().astype(np.float64)])
        sim.close()
    assert np.array_equal(outs["quad"], outs["lanes"], equal_nan=True)


@pytest.mark.parametrize("layout", ["epl", "lanes"])
def test_vss_ball_inside_a_resting_robot_is_not_a_contact(oracle_mod, monkeypatch, layout):
    """VSS-v0: the ball EXACTLY at the centre of the agent's robot, which is fed zero actions and stays put — the pair has no
    contact normal and the model skips it (0 < d2 < thr) in every sub-step.  The one-lane-per-env kernel finds pairs with one
    float compare (d2 < thr) and applies 0 < d2 where a pair is walked (the dropped pair must not count as touched either:
    a sum of +0 added to a velocity of -0 would change its bits); other robots pile onto the same spot to make real contacts."""
    import torch
    L = _lib()
    O = oracle_mod
    kind, ft, nb, ny = 0, 0, 3, 3
    B = 70
    monkeypatch.setenv("RSX_LAYOUT", layout)
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(1, 3, 0, 0)
    rng = np.random.default_rng(11)
    ball = np.zeros((B, 4)); rob = np.zeros((B, 6, 3))
    for e in range(B):
        rob[e, :, 0] = rng.uniform(-0.6, 0.6, 6); rob[e, :, 1] = rng.uniform(-0.5, 0.5, 6)
        rob[e, :, 2] = rng.uniform(-180, 180, 6)
        ball[e, :2] = rob[e, 0, :2]                                # the ball in the agent's place
        if e % 2: rob[e, 1 + e % 5, :2] = rob[e, 0, :2] + (0.05, 0.02)   # a neighbour overlapping robot 0 (and the ball)
    rob = np.float32(rob).astype(np.float64); ball[:, :2] = rob[:, 0, :2]   # exactly representable: the float state holds the same bits
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(1, 3, e, 0)
        r.task_reset()
        r.task_reset_to(ball[e], rob[e, :3], rob[e, 3:])
    sim.task_reset()
    sim.task_reset_to(ball, rob[:, :3], rob[:, 3:])
    tens = sim.task_tensors()
    a = np.zeros(tuple(tens["actions"].shape), np.float32)
    for t in range(10):
        if t >= 5:
            a = rng.uniform(-1, 1, a.shape).astype(np.float32)
        tens["actions"].copy_(torch.from_numpy(a))
        sim.task_step(tens["actions"].data_ptr())
        for e, r in enumerate(refs):
            r.task_step(a[e])
        _cmp_task(sim, refs, tens, t)
    sim.close()


@pytest.mark.parametrize("layout", ["quad", "lanes"])
def test_scrimmage_robots_in_one_place_are_not_a_contact(oracle_mod, monkeypatch, layout):
    """Two robots at EXACTLY the same position have no contact normal: the model skips the pair (0 < d2 < (2 r)^2).  The
    four-lane kernel finds pairs with a float compare (d2 < thr) and drops the zero distance where partners are walked;
    pairs inside one lane's six robots, across neighbouring lanes and across the diagonal, next to ordinary contacts."""
    import torch
    L = _lib()
    O = oracle_mod
    kind, ft, nb, ny = 1, 1, 11, 11
    B = 21
    monkeypatch.setenv("RSX_LAYOUT", layout)
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(6, 9, 0, 0)
    rng = np.random.default_rng(5)
    gx, gy = np.meshgrid(np.linspace(-2.5, 2.5, 6), np.linspace(-1.5, 1.5, 4))
    ball = np.zeros((B, 4)); rob = np.zeros((B, 22, 3))
    same = [(0, 1), (2, 7), (3, 15), (20, 21), (5, 18), (11, 12), (4, 10)]
    for e in range(B):
        rob[e, :, :2] = np.stack([gx.ravel(), gy.ravel()], 1)[:22] + rng.uniform(-0.05, 0.05, (22, 2))
        rob[e, :, 2] = rng.uniform(-180, 180, 22)
        a, b = same[e % len(same)]
        rob[e, b, :2] = rob[e, a, :2]                       # exactly coincident
        c = (a + 2) % 22 if (a + 2) % 22 != b else (a + 3) % 22
        rob[e, c, :2] = rob[e, a, :2] + (0.12, 0.05)        # and a third robot overlapping both
        ball[e, :2] = (3.5, 2.5)
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(6, 9, e, 0)
        r.task_reset()
        r.task_reset_to(ball[e], rob[e, :11], rob[e, 11:])
    sim.task_reset()
    sim.task_reset_to(ball, rob[:, :11], rob[:, 11:])
    tens = sim.task_tensors()
    a = np.zeros(tuple(tens["actions"].shape), np.float32)   # nobody drives: the coincident pairs stay coincident unless pushed
    for t in range(12):
        if t >= 6:
            a = rng.uniform(-1, 1, a.shape).astype(np.float32)
        tens["actions"].copy_(torch.from_numpy(a))
        sim.task_step(tens["actions"].data_ptr())
        for e, r in enumerate(refs):
            r.task_step(a[e])
        _cmp_task(sim, refs, tens, t)
    sim.close()


@pytest.mark.parametrize("B,n_step,n_roll", [(32768, 40, 0), (65536, 12, 18)], ids=["32768-steps", "65536-steps-and-a-multi-step-call"])
def test_scrimmage_large_batch_switches_to_the_quad_layout_and_agrees(monkeypatch, B, n_step, n_roll):
    """from 32 768 envs the spread scrimmage task picks the four-lanes-per-env kernel by itself; forcing the 32-lane kernel on
    the same seeds gives the same buffers (full size, resets included).  From 49 152 envs a multi-step call
    (rsx_task_rollout) on such a handle is issued as single-step launches of that kernel: same results as the 32-lane
    kernel's one launch."""
    import torch
    L = _lib()
    outs = []
    for layout in (None, "lanes"):
        if layout:
            monkeypatch.setenv("RSX_LAYOUT", layout)
        else:
            monkeypatch.delenv("RSX_LAYOUT", raising=False)
        sim = L.Sim(1, 1, 11, 11, 25, B)
        sim.task_attach(6, 2025, 0, 25)
        tens = sim.task_tensors()
        sim.task_reset()
        sim.task_step_n(n_step)
        if n_roll:
            sim.task_rollout(n_roll)
        torch.cuda.synchronize()
        outs.append((tens["obs"].clone(), tens["reward"].clone(), sim.state_tensor().clone(), sim.read_metrics()))
        sim.close()
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    assert np.array_equal(outs[0][3], outs[1][3]) and outs[0][3][1] >= B


@pytest.mark.parametrize("task,kind,ft,nb,ny,B,steps,layout", [(1, 0, 0, 3, 3, 512, 3000, None), (2, 1, 2, 1, 6, 384, 2000, None),
                                                              (5, 1, 2, 2, 0, 256, 1500, None), (1, 0, 0, 3, 3, 448, 2500, "epl"),
                                                              (6, 1, 1, 11, 11, 72, 1500, "quad"), (7, 1, 1, 11, 11, 40, 1500, "quad")],
                         ids=["vss", "static-defenders", "pass-endurance", "vss-one-lane", "scrimmage-four-lanes", "scrimmage-crowded-four-lanes"])
def test_long_horizon_bitexact(oracle_mod, monkeypatch, task, kind, ft, nb, ny, B, steps, layout):
    """Millions of env-steps against the oracle (OpenMP over envs): thousands of contacts, kicks,
    resets and TimeLimit truncations; the state is compared every 500 steps — any divergence in a
    chaotic system persists, so agreement at the checkpoints means agreement throughout.  The large-batch kernels
    (one lane per env, four lanes per env) are forced on small batches for their cases."""
    import torch
    L = _lib()
    O = oracle_mod
    O.set_threads(min(16, os.cpu_count() or 1))
    seed = 20260928
    if layout:
        monkeypatch.setenv("RSX_LAYOUT", layout)
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(task, seed, 0, 0)
    tens = sim.task_tensors()
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(task, seed, e, 0)
        r.task_reset()
    sim.task_reset()
    done = 0
    while done < steps:
        n = min(500, steps - done)
        sim.task_step_n(n - 1)
        sim.task_step(None)
        O.vec_task_step(refs, n)
        done += n
        torch.cuda.synchronize()
        st = sim.get_state_full()
        want = np.stack([r.get_state_full() for r in refs])
        assert f32_equal(st, want), mismatch_report(st, want, f"state after {done} steps")
    got = sim.read_metrics()
    want = sum(r.task_out()["metrics"] for r in refs)
    assert np.array_equal(got, want), (got, want)
    assert got[1] > B // 4
    sim.close()


@pytest.mark.parametrize("task,kind,ft,nb,ny,adim", EPL_TASKS[1:], ids=EPL_IDS[1:])
def test_ssl_env_per_lane_long_run_with_contacts(oracle_mod, monkeypatch, task, kind, ft, nb, ny, adim):
    """the SSL one-lane-per-env kernels over 1500 random-action steps x 96 envs (ball carried, kicked into the
    other robots, robot-robot hits, every termination branch): still the oracle's bits"""
    L = _lib()
    monkeypatch.setenv("RSX_LAYOUT", "epl")
    B = 96
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(task, 99, 0, 0)
    tens = sim.task_tensors()
    sim.task_reset()
    refs = _mk_oracles(oracle_mod, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(task, 99, e, 0)
        r.task_reset()
    for chunk in range(6):
        sim.task_step_n(249)
        sim.task_step(None)
        oracle_mod.vec_task_step(refs, 250)
        _cmp_task(sim, refs, tens, chunk)
    m = sim.read_metrics()
    assert np.array_equal(m, sum(r.task_out()["metrics"] for r in refs)) and m[1] >= B
    sim.close()


@pytest.mark.parametrize("task,kind,ft,nb,ny,adim", EPL_TASKS, ids=EPL_IDS)
def test_large_batch_switches_layout_and_agrees(monkeypatch, task, kind, ft, nb, ny, adim):
    """At 131 072 envs (above every layout threshold) the library picks the one-lane-per-env kernel by itself; forcing the 8-lane
    kernel on the same seeds must give the same buffers (full size, a few hundred resets)."""
    import torch
    L = _lib()
    B = 131072
    outs = []
    for layout in (None, "lanes"):
        if layout:
            monkeypatch.setenv("RSX_LAYOUT", layout)
        else:
            monkeypatch.delenv("RSX_LAYOUT", raising=False)
        sim = L.Sim(kind, ft, nb, ny, 25, B)
        sim.task_attach(task, 2025, 0, 30 if task >= 6 else 0)   # (scrimmage episodes are long: end them by the step limit)
        tens = sim.task_tensors()
        sim.task_reset()
        sim.task_step_n(25)
        sim.task_rollout(15)
        torch.cuda.synchronize()
        outs.append((tens["obs"].clone(), tens["reward"].clone(), sim.state_tensor().clone(), sim.read_metrics()))
        sim.close()
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    assert np.array_equal(outs[0][3], outs[1][3]) and outs[0][3][1] > 0


def test_api_errors_are_reported_not_crashed():
    L = _lib()
    sim = L.Sim(0, 0, 3, 3, 25, 8)
    with pytest.raises(L.RsxError, match="no task attached"):
        sim.task_step(None)
    with pytest.raises(L.RsxError, match="does not match"):
        sim.task_attach(2, 0, 0, 0)           # StaticDefenders needs an SSL handle
    sim.task_attach(1, 0, 0, 0)
    with pytest.raises(L.RsxError, match="already attached"):
        sim.task_attach(1, 0, 0, 0)
    for call in (lambda: sim.task_step(None), lambda: sim.task_step_n(3), lambda: sim.task_rollout(3)):
        with pytest.raises(L.RsxError, match="must come before the first step"):
            call()                            # no episode is open yet
    sim.task_reset()
    with pytest.raises(L.RsxError):
        sim.task_rollout(-1)
    sim.task_step(None)
    sim.close()
    for bad in ((2, 0, 3, 3), (0, 7, 3, 3), (0, 0, 0, 0), (0, 0, 20, 20)):
        with pytest.raises(L.RsxError, match="bad simulator configuration"):
            L.Sim(*bad, 25, 4)
    with pytest.raises(L.RsxError, match="device_id"):
        L.Sim(0, 0, 3, 3, 25, 4, device_id=99)
    ssl = L.Sim(1, 2, 1, 6, 25, 4)
    with pytest.raises(L.RsxError, match="does not match"):
        ssl.task_attach(3, 0, 0, 0)           # dribbling needs 1v4
    ssl.close()
    # rows are addressed with 32-bit byte offsets: a batch whose state array would reach 4 GB is refused before anything is allocated
    with pytest.raises(L.RsxError, match="4 GB"):
        L.Sim(1, 1, 11, 11, 25, 5_000_000)    # 11v11: 249 rows x 5 M envs x 4 B
    # the two 32-bit words of the Philox counter that a caller can overrun are range-checked, not wrapped (rsx.h)
    sim = L.Sim(0, 0, 3, 3, 25, 8)
    with pytest.raises(L.RsxError, match=r"exceeds 2\^32"):
        sim.task_attach(1, 0, 2 ** 32 - 4, 0)            # global env ids 2^32 - 4 .. 2^32 + 3
    with pytest.raises(L.RsxError, match=r"exceeds 2\^32"):
        sim.task_attach(1, 0, 2 ** 40, 0)
    sim.task_attach(1, 0, 2 ** 32 - 8, 0)                # the last eight ids are fine
    sim.task_reset()
    sim.task_step_n(3)
    blob = sim.task_checkpoint().copy()
    assert int(blob[76:80].view(np.uint32)[0]) == 3      # the handle's step counter in the checkpoint header
    blob[76:80] = np.array([0xFFFFFFFD], dtype=np.uint32).view(np.uint8)
    sim.task_restore(blob)
    before = sim.get_state_full()
    for call in (lambda: sim.task_step_n(3), lambda: sim.task_rollout(3)):
        with pytest.raises(L.RsxError, match="step counter exhausted"):
            call()                                       # 2^32 - 3 + 3 steps would wrap the counter
    assert np.array_equal(before, sim.get_state_full())  # a refused call changes nothing
    sim.task_step_n(2)                                   # the last two steps a handle can take
    with pytest.raises(L.RsxError, match="step counter exhausted"):
        sim.task_step(None)
    sim.close()


@pytest.mark.parametrize("task,kind,ft,nb,ny,max_steps", [(2, 1, 2, 1, 6, 9), (2, 1, 2, 1, 6, 6), (2, 1, 2, 1, 6, 1)])
def test_placement_cache_serves_resets_bit_identically(oracle_mod, monkeypatch, task, kind, ft, nb, ny, max_steps):
    """Latency-bound batches: helper workgroups of every step launch compute each env's next placement ahead of time
    (rsx_kernels.hpp: placement_helper) and the resetting wave copies it.  Same poses as the inline placement — the run
    equals the oracle's bit for bit — and the cache really is what served them (counters); an episode length of one
    step (a reset in every launch: the entry is never ready in time) falls back to the inline path, also bit-exact."""
    import torch
    L = _lib()
    monkeypatch.setenv("RSX_PCACHE_STATS", "1")
    B, seed, base, steps = 37, 77, 4000, 64
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(task, seed, base, max_steps)
    assert sim.placement_cache_stats() == (0, 0)
    tens = sim.task_tensors()
    refs = _mk_oracles(oracle_mod, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(task, seed, base + e, max_steps)
        r.task_reset()
    sim.task_reset()
    for t in range(steps):
        sim.task_step(None)
        for r in refs:
            r.task_step(None)
        if t % 7 == 6 or t == steps - 1:
            _cmp_task(sim, refs, tens, t)
    hits, inline = sim.placement_cache_stats()
    episodes = int(sim.read_metrics()[1])
    assert hits + inline == episodes and episodes >= B * (steps // max_steps) // 2
    if max_steps == 1:
        assert inline > 0                      # a reset in every launch: (nearly) every placement is computed in place
    else:
        assert hits >= 0.9 * episodes          # after the first two launches every reset finds its poses ready
    sim.close()
    monkeypatch.setenv("RSX_NO_PCACHE", "1")
    off = L.Sim(kind, ft, nb, ny, 25, B)
    off.task_attach(task, seed, base, max_steps)
    assert off.placement_cache_stats() == (-1, -1)
    off.close()


def test_api_calls_leave_the_current_device_alone():
    """No C-ABI call may change the thread's current HIP device (= torch's current device), also not
    the destructor run by the garbage collector.  With a single device this pins the common case;
    with two, a handle on the other device is created, stepped and destroyed while device 0 is current."""
    import gc
    import torch
    L = _lib()
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(0)
    other = 1 if ndev >= 2 else 0
    sim = L.Sim(0, 0, 3, 3, 25, 64, device_id=other)
    assert torch.cuda.current_device() == 0
    sim.task_attach(1, 3, 0, 0)
    sim.task_reset()
    sim.task_step_n(5)
    st = sim.get_state()
    m = sim.read_metrics()
    assert torch.cuda.current_device() == 0 and m[0] == 5 * 64 and np.all(np.isfinite(st))
    x = torch.zeros(4, device="cuda")          # lands on torch's current device, not the handle's
    assert x.device.index == 0
    del sim
    gc.collect()
    assert torch.cuda.current_device() == 0


def test_finite_check_and_debug_mode():
    """rsx_check_finite counts non-finite floats; RSX_DEBUG_FINITE=1 turns every stepping call into a
    checked one (RSX_ERR_STATE).  A NaN is planted through set_state."""
    import subprocess, sys
    L = _lib()
    sim = L.Sim(0, 0, 3, 3, 25, 16)
    sim.task_attach(1, 0, 0, 0)
    sim.task_reset()
    sim.task_step_n(20)
    assert sim.check_finite() == 0
    st = sim.get_state_full()
    st[3, 5] = np.nan; st[7, 0] = np.inf
    sim.set_state(st)
    assert sim.check_finite() == 2
    sim.close()
    code = ("import numpy as np\nfrom rsoccer_amd import _lib as L\n"
            "s = L.Sim(0, 0, 3, 3, 25, 16); s.task_attach(1, 0, 0, 0); s.task_reset(); s.task_step_n(5)\n"
            "st = s.get_state_full(); st[3, 5] = np.nan; s.set_state(st)\n"
            "try:\n    s.task_step(None)\n    print('NOT CAUGHT')\n"
            "except L.RsxError as e:\n    print('CAUGHT', e)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root,
                         env=dict(os.environ, RSX_DEBUG_FINITE="1", PYTHONPATH=root), timeout=300)
    assert "CAUGHT" in res.stdout and "RSX_DEBUG_FINITE" in res.stdout and "NOT CAUGHT" not in res.stdout, res.stdout + res.stderr


def test_metrics_env_steps_are_counted_on_the_device():
    """metrics[0] is part of the device vector (what the multi-GPU all-reduce sends), for single-step
    launches, C-side step loops and one-launch rollouts alike."""
    L = _lib()
    sim = L.Sim(0, 0, 3, 3, 25, 100)
    sim.task_attach(1, 1, 0, 5)
    sim.task_reset()
    sim.task_step(None); sim.task_step_n(4); sim.task_rollout(7)
    import torch
    torch.cuda.synchronize()
    dev = sim.task_tensors()["metrics"].cpu().numpy()
    assert dev[0] == 100 * 12
    # the episode counters are per-block partial sums until they are folded (rsx_metrics_fold)
    sim.metrics_fold()
    torch.cuda.synchronize()
    dev = sim.task_tensors()["metrics"].cpu().numpy()
    assert dev[1] == 200 and dev[5] == 1000 and dev[6] == 200 and np.array_equal(dev, sim.read_metrics())
    sim.task_rollout(5)
    assert sim.read_metrics()[1] == 300 and sim.read_metrics()[1] == 300   # folding twice adds nothing
    sim.close()


@pytest.mark.parametrize("ts_ms", [0, 3, 12, 50])
def test_other_time_steps_bitexact(oracle_mod, ts_ms):
    """sub-step count follows time_step_ms (0 = no physics: commands, observation, reward only)"""
    L = _lib()
    O = oracle_mod
    B = 11
    for task, kind, ft, nb, ny in ((1, 0, 0, 3, 3), (2, 1, 2, 1, 6)):
        sim = L.Sim(kind, ft, nb, ny, ts_ms, B)
        sim.task_attach(task, 17, 0, 25)
        tens = sim.task_tensors()
        refs = [O.OracleEnv(kind, ft, nb, ny, ts_ms, "f32") for _ in range(B)]
        for e, r in enumerate(refs):
            r.task_attach(task, 17, e, 25)
            r.task_reset()
        sim.task_reset()
        for t in range(60):
            sim.task_step(None)
            for r in refs:
                r.task_step(None)
        _cmp_task(sim, refs, tens, f"ts={ts_ms}")
        sim.close()


@pytest.mark.parametrize("layout", ["lanes", "quad"])
def test_scrimmage_wall_and_goal_contacts_in_both_layouts(oracle_mod, monkeypatch, layout):
    """The SSL kernels skip the wall clamp in waves with no body near a wall (near_walls, rsx_body.hpp).  Here the bodies
    START at the walls: robots lined up along the touch lines, the goal lines, inside the goal mouths and behind the
    goals, the ball in a corner / a goal / mid-field, fed commands that push them outwards — the CPU oracle's bits for
    80 steps in the 32-lane and the four-lane kernel (ragged batch: envs without any body near a wall share waves
    with envs full of them)."""
    import torch
    L = _lib()
    O = oracle_mod
    monkeypatch.setenv("RSX_LAYOUT", layout)
    kind, ft, nb, ny, B = 1, 1, 11, 11, 37
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(6, 5, 0, 0)
    f = sim.get_field_params()
    hl, hw, gd, ghw = f["length"] / 2, f["width"] / 2, f["goal_depth"], f["goal_width"] / 2
    rng = np.random.default_rng(8)
    ball = np.zeros((B, 4)); rob = np.zeros((B, 22, 3))
    for e in range(B):
        if e % 3 == 2:      # an env far from every wall
            gx, gy = np.meshgrid(np.linspace(-2.5, 2.5, 6), np.linspace(-1.5, 1.5, 4))
            rob[e, :, :2] = np.stack([gx.ravel(), gy.ravel()], 1)[:22]
            ball[e, :2] = (0.3, 0.2)
        else:
            for k in range(22):
                side = 1.0 if k % 2 else -1.0
                if k < 8:    rob[e, k, :2] = (-4.5 + 1.2 * k, side * (hw + 0.2))                 # along the touch lines (inside the margin)
                elif k < 14: rob[e, k, :2] = (side * (hl + 0.15), -3.0 + 1.1 * (k - 8) + 1.2)      # on the goal lines, outside the mouth
                elif k < 18: rob[e, k, :2] = (side * (hl + 0.05 * (k - 13)), (k - 15.5) * 0.25)   # in the goal mouths
                else:        rob[e, k, :2] = (side * (hl + gd + 0.12), (k - 19.5) * 0.8)          # behind the goals
            ball[e] = [(hl + 0.1, ghw - 0.03, 1.0, 1.5), (hl - 0.02, hw - 0.02, 2.0, 2.0)][e % 2] if e % 3 == 0 else (-(hl + gd - 0.03), 0.1, -3.0, 0.4)
        rob[e, :, 2] = rng.uniform(-180, 180, 22)
        rob[e, :, :2] += rng.uniform(-0.01, 0.01, (22, 2))
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(6, 5, e, 0)
        r.task_reset()
        r.task_reset_to(ball[e], rob[e, :11], rob[e, 11:])
    sim.task_reset()
    sim.task_reset_to(ball, rob[:, :11], rob[:, 11:])
    tens = sim.task_tensors()
    for t in range(80):
        a = rng.uniform(-1, 1, tuple(tens["actions"].shape)).astype(np.float32)
        a.reshape(B, 22, 4)[:, :, 0] = np.abs(a.reshape(B, 22, 4)[:, :, 0])      # keep driving forwards (headings are random: half of them outwards)
        tens["actions"].copy_(torch.from_numpy(a))
        sim.task_step(tens["actions"].data_ptr())
        for e, r in enumerate(refs):
            r.task_step(a[e])
        if t % 8 == 7:
            _cmp_task(sim, refs, tens, t)
    st = sim.get_state_full()
    assert np.abs(st[:, 5::11][:, :22]).max() <= hl + gd + 0.35 + 1e-4     # nobody left the playable region
    sim.close()


@pytest.mark.parametrize("layout", ["lanes", "epl"])
def test_static_defenders_wall_contacts_in_both_layouts(oracle_mod, monkeypatch, layout):
    """The same for SSLStaticDefenders-v0 (8 lanes per env / one lane per env): the agent driven into the walls and the
    goal, defenders parked on the walls, the ball shot into corners, along walls and into the goal mouth."""
    import torch
    L = _lib()
    O = oracle_mod
    monkeypatch.setenv("RSX_LAYOUT", layout)
    kind, ft, nb, ny, B = 1, 2, 1, 6, 70
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(2, 9, 0, 0)
    f = sim.get_field_params()
    hl, hw, gd, ghw = f["length"] / 2, f["width"] / 2, f["goal_depth"], f["goal_width"] / 2
    rng = np.random.default_rng(12)
    ball = np.zeros((B, 4)); rob = np.zeros((B, 7, 3))
    for e in range(B):
        if e % 4 == 3:      # far from every wall
            rob[e, :, :2] = [(-1.0 + 0.4 * k, -0.6 + 0.2 * k) for k in range(7)]
            ball[e] = (0.2, 0.5, 0.5, -0.5)
        else:
            sx = 1.0 if e % 2 else -1.0
            rob[e, 0, :2] = [(sx * (hl + 0.1), 0.1), (sx * (hl - 0.3), hw + 0.15), (sx * (hl + gd - 0.02), -0.2)][e % 3]
            for k in range(1, 7):
                rob[e, k, :2] = [(sx * (hl + 0.2), -1.6 + 0.55 * k), (-2.0 + 0.7 * k, -sx * (hw + 0.2))][k % 2]
            ball[e] = [(sx * (hl - 0.05), hw - 0.05, sx * 2.5, 2.0), (sx * (hl + 0.05), ghw - 0.04, sx * 1.0, 1.0), (0.0, -(hw + 0.2), 1.0, -2.0)][e % 3]
        rob[e, :, 2] = rng.uniform(-180, 180, 7)
        rob[e, :, :2] += rng.uniform(-0.01, 0.01, (7, 2))
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    for e, r in enumerate(refs):
        r.task_attach(2, 9, e, 0)
        r.task_reset()
        r.task_reset_to(ball[e], rob[e, :1], rob[e, 1:])
    sim.task_reset()
    sim.task_reset_to(ball, rob[:, :1], rob[:, 1:])
    tens = sim.task_tensors()
    for t in range(60):
        a = rng.uniform(-1, 1, tuple(tens["actions"].shape)).astype(np.float32)
        a[:, 0] = np.abs(a[:, 0])
        tens["actions"].copy_(torch.from_numpy(a))
        sim.task_step(tens["actions"].data_ptr())
        for e, r in enumerate(refs):
            r.task_step(a[e])
        if t % 6 == 5:
            _cmp_task(sim, refs, tens, t)
    sim.close()


@pytest.mark.parametrize("task", [6, 7], ids=["spread", "crowded"])
def test_scrimmage_checkpoint_resume_across_the_quad_and_lane_layouts(monkeypatch, task):
    """The 11v11 task: a run saved under the four-lanes-per-env kernel continues under the 32-lane kernel (and the other
    way round) exactly as the uninterrupted run — ragged batch, TimeLimit resets, single-step and multi-step calls."""
    import torch
    L = _lib()
    B, seed, base = 83, 99, 5

    def snapshot(sim, tens):
        torch.cuda.synchronize()
        return np.concatenate([sim.get_state_full().ravel()] + [tens[k].cpu().numpy().astype(np.float64).ravel()
                              for k in ("obs", "reward", "terminated", "truncated", "final_obs", "steps")]
                              + [sim.read_metrics().astype(np.float64)])

    def make(layout):
        monkeypatch.setenv("RSX_LAYOUT", layout)
        sim = L.Sim(1, 1, 11, 11, 25, B)
        sim.task_attach(task, seed, base, 30)
        return sim, sim.task_tensors()

    for first, second in (("quad", "lanes"), ("lanes", "quad")):
        a, ta = make(first)
        a.task_reset()
        a.task_step_n(41); a.task_rollout(7); a.task_step_n(2)
        blob = a.task_checkpoint()
        a.task_step_n(25); a.task_rollout(9); a.task_step(None)
        want = snapshot(a, ta)
        a.close()
        b, tb = make(second)
        b.task_restore(blob)
        b.task_step_n(25); b.task_rollout(9); b.task_step(None)
        assert np.array_equal(snapshot(b, tb), want, equal_nan=True), (first, second)
        b.close()


@pytest.mark.parametrize("task,kind,ft,nb,ny,adim", EPL_TASKS, ids=EPL_IDS)
def test_checkpoint_resume_is_bit_identical_across_handles_and_layouts(monkeypatch, task, kind, ft, nb, ny, adim):
    """rsx_task_checkpoint_save / _load: a run interrupted after 70 steps continues in a NEW handle (stepped by the
    other kernel layout) exactly as the uninterrupted one: state, observations, rewards, flags, episode bookkeeping,
    random streams and metrics."""
    import torch
    L = _lib()
    B, seed, base = 200, 4242, 31

    def snapshot(sim, tens):
        torch.cuda.synchronize()
        return np.concatenate([sim.get_state_full().ravel()] + [tens[k].cpu().numpy().astype(np.float64).ravel()
                              for k in ("obs", "reward", "terminated", "truncated", "info", "final_obs", "steps")]
                              + [sim.read_metrics().astype(np.float64)])

    def make(layout):
        monkeypatch.setenv("RSX_LAYOUT", layout)
        sim = L.Sim(kind, ft, nb, ny, 25, B)
        sim.task_attach(task, seed, base, 40)
        return sim, sim.task_tensors()

    for first, second in (("lanes", "epl"), ("epl", "lanes")):   # saved by one kernel layout, continued by the other
        a, ta = make(first)
        a.task_reset()
        a.task_step_n(50); a.task_rollout(20)
        if first == "epl":
            a.task_step_n(3)   # the last launch before the save is a single-step one (the kernel that does not keep every aux row current)
        blob = a.task_checkpoint()
        assert blob.dtype == np.uint8 and blob.size > B * 4 * (5 + 6 * (nb + ny))
        a.task_step_n(30); a.task_rollout(25); a.task_step(None)
        want = snapshot(a, ta)
        a.close()

        b, tb = make(second)                  # another handle, the other layout, never reset
        with pytest.raises(L.RsxError, match="must come before the first step"):
            b.task_step(None)
        b.task_restore(blob)
        b.task_step_n(30); b.task_rollout(25); b.task_step(None)
        got = snapshot(b, tb)
        assert np.array_equal(got, want, equal_nan=True), (first, second)
        b.close()

    c, _ = make("lanes")                      # a damaged blob is refused
    with pytest.raises(L.RsxError, match="truncated"):
        c.task_restore(blob[: blob.size // 2])
    c.close()
    d = L.Sim(kind, ft, nb, ny, 25, B)        # so is another seed ...
    d.task_attach(task, seed + 1, base, 40)
    with pytest.raises(L.RsxError, match="another seed"):
        d.task_restore(blob)
    d.close()
    e = L.Sim(kind, ft, nb, ny, 25, B + 1)   # ... and another batch size
    e.task_attach(task, seed, base, 40)
    with pytest.raises(L.RsxError, match="different configuration"):
        e.task_restore(blob)
    e.close()
    f = L.Sim(kind, ft, nb, ny, 20, B)       # ... another time step
    f.task_attach(task, seed, base, 40)
    with pytest.raises(L.RsxError, match="field type or time step"):
        f.task_restore(blob)
    f.close()
    g = L.Sim(kind, ft, nb, ny, 25, B)       # ... and another TimeLimit
    g.task_attach(task, seed, base, 41)
    with pytest.raises(L.RsxError, match="max_episode_steps"):
        g.task_restore(blob)
    g.close()
    m = L.Sim(kind, ft, nb, ny, 25, B)       # ... and a blob saved under another version of the physics model (header word `model`:
    m.task_attach(task, seed, base, 40)      # its trajectories would not continue bit-identically under this one)
    other = np.array(blob, copy=True)
    off = 8 + 4 * 13                         # magic, then ten int32 + field_type, time_step_ms, max_steps
    assert int(other[off:off + 4].view(np.int32)[0]) == 2    # RSX_PHYSICS_MODEL
    other[off:off + 4] = np.array([1], dtype=np.int32).view(np.uint8)
    with pytest.raises(L.RsxError, match="physics model"):
        m.task_restore(other)
    # ... and a header whose section sizes (used as copy lengths) disagree with its configuration: a damaged file
    for word in range(4):                    # state_bytes, aux_bytes, obs_bytes, flag_bytes: uint64 at 80, 88, 96, 104
        bad = np.array(blob, copy=True)
        bad[80 + 8 * word: 88 + 8 * word].view(np.uint64)[0] += np.uint64(4096)
        with pytest.raises(L.RsxError, match="damaged"):
            m.task_restore(np.concatenate([bad, np.zeros(8192, dtype=np.uint8)]))
    m.close()


def _pile_at_walls(rng, B, N, hl, hw, ghw, gd, r, margin, vss):
    """B placements of N robots as piles pressed against a boundary wall, into a corner, into a goal or around a goal post:
    rows of robots that overlap their neighbours by 8 mm (deeper than the second-sweep threshold), the outermost one standing
    1 mm inside the wall.  Returns (ball [B,4], robots [B,N,3], where each pile sits)."""
    ball = np.zeros((B, 4)); rob = np.zeros((B, N, 3))
    xl, yl = hl + margin - r, hw + margin - r
    step = 2 * r - 0.008
    for e in range(B):
        kind = e % 4
        sx, sy = (1.0 if (e // 4) % 2 else -1.0), (1.0 if (e // 8) % 2 else -1.0)
        for k in range(N):
            row, col = k // 3, k % 3
            if kind == 0:      # rows perpendicular to a side wall (|y| = yl), the first of each row at the wall
                rob[e, k, :2] = (sx * (0.3 + row * (2 * r + 0.01)), sy * (yl - 0.001 - col * step))
            elif kind == 1:    # into a corner: both axes blocked for the first robot
                rob[e, k, :2] = (sx * (xl - 0.001 - row * step), sy * (yl - 0.001 - col * step))
            elif kind == 2:    # VSS: into the goal box / SSL: against the end wall beside the goal
                if vss:
                    rob[e, k, :2] = (sx * (hl + gd - r - 0.001 - col * step), sy * (row * (2 * r + 0.004) - 0.02))
                else:
                    rob[e, k, :2] = (sx * (xl - 0.001 - col * step), sy * (ghw + 0.4 + row * (2 * r + 0.01)))
            else:              # around a goal post: a robot 1 cm from the post point, neighbours overlapping it
                rob[e, k, :2] = (sx * (hl - 0.01 - col * step * 0.7 + 0.03 * row), sy * (ghw + (r - 0.02) * (1 - 2 * (row % 2)) + 0.05 * col - row * 0.16))
        rob[e, :, 2] = rng.uniform(-180, 180, N)
        ball[e, :2] = (sx * 0.1, -sy * 0.2)
    return ball, rob


def _assert_wall_pairs_seen(st, N, rs, r, xl, yl):
    """coverage: some env holds a touching robot pair one of whose robots stands at a boundary wall"""
    x, y = st[:, 5::rs][:, :N], st[:, 6::rs][:, :N]
    d = np.hypot(x[:, :, None] - x[:, None], y[:, :, None] - y[:, None]) + 9.0 * np.eye(N)[None]
    at_wall = (np.abs(x) > xl - 2e-3) | (np.abs(y) > yl - 2e-3)
    touching = (d < 2 * r) & (at_wall[:, :, None] | at_wall[:, None])
    assert touching.any()


@pytest.mark.parametrize("kind,ft,nb,ny,lanes", [(0, 0, 3, 3, None), (0, 1, 5, 5, None), (1, 2, 3, 2, None), (1, 0, 6, 6, None),
                                                 (1, 1, 11, 11, None), (0, 0, 3, 3, "64")],
                         ids=["vss-3v3", "vss-5v5", "ssl-3v2-run-time-count", "ssl-6v6", "ssl-11v11", "vss-3v3-one-wave-per-env"])
def test_piles_pressed_against_walls_raw_step_bitexact(oracle_mod, monkeypatch, kind, ft, nb, ny, lanes):
    """Model v2: wall-aware shares of robot-robot pairs, the third and fourth contact sweep of piles at a wall, goal
    posts — piles pushed into side walls, corners, goals and posts by their own drive, every lane-group width."""
    L = _lib()
    O = oracle_mod
    if lanes:
        monkeypatch.setenv("RSX_LANES_PER_ENV", lanes)
    B, N = 48, nb + ny
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    f = sim.get_field_params()
    hl, hw, ghw, gd, r = f["length"] / 2, f["width"] / 2, f["goal_width"] / 2, f["goal_depth"], f["rbt_radius"]
    margin = 0.3 if kind == 1 else 0.0
    rng = np.random.default_rng(77)
    ball, rob = _pile_at_walls(rng, B, N, hl, hw, ghw, gd, r, margin, kind == 0)
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    sim.reset(ball, rob[:, :nb], rob[:, nb:])
    for e in range(B):
        refs[e].reset(ball[e], rob[e, :nb], rob[e, nb:])
    C, rs = sim.cmd_dim, (6 if kind == 0 else 11)
    seen = False
    for t in range(50):
        st = sim.get_state()
        x, y, th = st[:, 5::rs][:, :N], st[:, 6::rs][:, :N], np.deg2rad(st[:, 7::rs][:, :N])
        # everybody drives outwards (away from the field centre): the piles stay pressed on their walls
        gx, gy = np.sign(x) * (1.0 + 0.3 * rng.uniform(-1, 1, x.shape)), np.sign(y) * (1.0 + 0.3 * rng.uniform(-1, 1, y.shape))
        cm = np.zeros((B, N, C))
        if kind == 0:
            fwd = gx * np.cos(th) + gy * np.sin(th)
            cm[:, :, 0] = 30.0 * fwd + rng.uniform(-8, 8, fwd.shape); cm[:, :, 1] = 30.0 * fwd + rng.uniform(-8, 8, fwd.shape)
        else:
            cm[:, :, 1] = 1.5 * (gx * np.cos(th) + gy * np.sin(th)); cm[:, :, 2] = 1.5 * (-gx * np.sin(th) + gy * np.cos(th))
            cm[:, :, 3] = rng.uniform(-2, 2, (B, N))
        sim.step(cm)
        for e in range(B):
            refs[e].step(cm[e])
        got = sim.get_state_full()
        for e in range(B):
            w = refs[e].get_state_full()
            assert f32_equal(got[e], w), mismatch_report(got[e], w, f"env {e} (pile kind {e % 4}) step {t}")
        if not seen:
            try:
                _assert_wall_pairs_seen(got, N, rs, r, hl + margin - r, hw + margin - r); seen = True
            except AssertionError:
                pass
    assert seen
    sim.close()


@pytest.mark.parametrize("task,layout,kind,ft,nb,ny", [(1, "lanes", 0, 0, 3, 3), (1, "epl", 0, 0, 3, 3), (2, "lanes", 1, 2, 1, 6), (2, "epl", 1, 2, 1, 6),
                                                         (7, "lanes", 1, 1, 11, 11), (7, "quad", 1, 1, 11, 11)],
                         ids=["vss-v0-lanes", "vss-v0-one-lane", "1v6-lanes", "1v6-one-lane", "11v11-lanes", "11v11-four-lanes"])
def test_piles_pressed_against_walls_fused_tasks_bitexact(oracle_mod, monkeypatch, task, layout, kind, ft, nb, ny):
    """The same piles under the fused task kernels of every layout (the large-batch kernels hold both bodies of a pair in one
    lane: contact_pair, rsx_body.hpp): episodes opened on the pile placements, the agents driven outwards."""
    import torch
    L = _lib()
    O = oracle_mod
    monkeypatch.setenv("RSX_LAYOUT", layout)
    B, N, seed = 40, nb + ny, 21
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(task, seed, 0, 0)
    assert sim.task_layout() == {"epl": "one-lane-per-env", "quad": "four-lanes-per-env"}.get(layout, sim.task_layout())
    f = sim.get_field_params()
    hl, hw, ghw, gd, r = f["length"] / 2, f["width"] / 2, f["goal_width"] / 2, f["goal_depth"], f["rbt_radius"]
    margin = 0.3 if kind == 1 else 0.0
    rng = np.random.default_rng(78)
    ball, rob = _pile_at_walls(rng, B, N, hl, hw, ghw, gd, r, margin, kind == 0)
    refs = _mk_oracles(O, kind, ft, nb, ny, B)
    for e, rf in enumerate(refs):
        rf.task_attach(task, seed, e, 0)
        rf.task_reset()
        rf.task_reset_to(ball[e], rob[e, :nb], rob[e, nb:])
    sim.task_reset()
    sim.task_reset_to(ball, rob[:, :nb], rob[:, nb:])
    tens = sim.task_tensors()
    rs = 6 if kind == 0 else 11
    _assert_wall_pairs_seen(sim.get_state_full(), N, rs, r, hl + margin - r, hw + margin - r)   # the first step starts from piles at walls
    seen = False
    for t in range(40):
        a = rng.uniform(-1, 1, tuple(tens["actions"].shape)).astype(np.float32)
        if task == 7:       # every robot is commanded: robot-local velocities that push outwards whatever the heading
            st = sim.get_state()
            x, y, th = st[:, 5::rs][:, :N], st[:, 6::rs][:, :N], np.deg2rad(st[:, 7::rs][:, :N])
            gx, gy = np.sign(x), np.sign(y)
            a = a.reshape(B, N, 4)
            a[:, :, 0] = 0.6 * (gx * np.cos(th) + gy * np.sin(th)); a[:, :, 1] = 0.6 * (-gx * np.sin(th) + gy * np.cos(th))
            a = np.ascontiguousarray(a.reshape(B, -1).astype(np.float32))
        tens["actions"].copy_(torch.from_numpy(a))
        sim.task_step(tens["actions"].data_ptr())
        for e, rf in enumerate(refs):
            rf.task_step(a[e])
        if t % 4 == 3 or t < 3:
            _cmp_task(sim, refs, tens, t)
            if not seen:
                try:
                    _assert_wall_pairs_seen(sim.get_state_full(), N, rs, r, hl + margin - r, hw + margin - r); seen = True
                except AssertionError:
                    pass
    assert seen or task == 2   # (1v6: the six defenders never drive — their piles are resolved once and stay apart)
    sim.close()


@pytest.mark.parametrize("kind,ft,nb,ny", [(0, 0, 3, 3), (1, 2, 1, 6)])
def test_host_format_step_with_and_without_zero_copy(oracle_mod, kind, ft, nb, ny):
    """rsx_step / rsx_step_state of a small handle read the commands from, and mirror the state into, pinned host memory (one
    launch, no copies); RSX_NO_ZERO_COPY=1 takes the copy path of the large handles.  Same results, bit for bit, also when host-
    format and device-resident steps alternate (the device command buffer is the caller's between them: rsx.h)."""
    import subprocess, sys, json
    child = r'''
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rsoccer_amd import _lib as L
kind, ft, nb, ny = map(int, sys.argv[1:5])
B, N = 5, nb + ny
sim = L.Sim(kind, ft, nb, ny, 25, B)
rng = np.random.default_rng(4)
out = []
for t in range(30):
    cm = rng.uniform(-1, 1, (B, N, sim.cmd_dim)) * (30.0 if kind == 0 else 1.5)
    if kind == 1: cm[:, :, 0] = 0; cm[:, :, 4:] = 0
    if t % 3 == 2:      # a device-resident step in between: the caller fills the command buffer
        sim.cmds_tensor().copy_(torch.from_numpy(cm.transpose(1, 2, 0).reshape(N * sim.cmd_dim, B).astype(np.float32)))
        sim.step_dev(); out.append(sim.get_state_full().tolist())
    else:
        out.append(sim.step_state(np.ascontiguousarray(cm)).tolist())
print(json.dumps(out))
'''
    res = []
    for env in ({}, {"RSX_NO_ZERO_COPY": "1"}):
        r = subprocess.run([sys.executable, "-c", child, str(kind), str(ft), str(nb), str(ny)], env=dict(os.environ, **env),
                           capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1]
    # and the oracle agrees with the host-format steps
    O = oracle_mod
    rng = np.random.default_rng(4)
    refs = _mk_oracles(O, kind, ft, nb, ny, 5)
    N = nb + ny
    cd = 2 if kind == 0 else 8
    for t in range(30):
        cm = rng.uniform(-1, 1, (5, N, cd)) * (30.0 if kind == 0 else 1.5)
        if kind == 1:
            cm[:, :, 0] = 0; cm[:, :, 4:] = 0
        cm32 = cm.astype(np.float32).astype(np.float64)
        for e in range(5):
            refs[e].step(cm32[e] if t % 3 == 2 else cm[e])
        got = np.array(res[0][t])
        for e in range(5):
            w = refs[e].get_state_full() if t % 3 == 2 else refs[e].get_state()
            assert f32_equal(got[e][:len(w)], w), mismatch_report(got[e][:len(w)], w, f"env {e} step {t}")


@pytest.mark.parametrize("kind,ft,nb,ny,B", [(0, 0, 3, 3, 4096), (1, 2, 1, 6, 1000)], ids=["VSS-3v3-4096", "SSL-1v6-1000"])
def test_wire_format_step_of_large_batches_converts_on_the_device(kind, ft, nb, ny, B):
    """Handles of more than 64 envs keep the reference's float64 wire format in pinned host buffers and convert on the device (ABI 6:
    rsx_wire_buffers / rsx_step_wire; rsx_step / rsx_step_state / rsx_get_state are a memcpy around the same path).  Bit-identical to
    the staging path they replace (RSX_NO_WIRE_PATH=1: transposing loop on the host + two copies) and to the device-resident step fed
    the same commands as float32; the wire state buffer holds get_state() + the two internal rows."""
    import subprocess, sys, hashlib
    child = r'''
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rsoccer_amd import _lib as L
kind, ft, nb, ny, B = map(int, sys.argv[1:6])
N = nb + ny
sim = L.Sim(kind, ft, nb, ny, 25, B)
ref = L.Sim(kind, ft, nb, ny, 25, B)          # device-resident twin
wire = sim.wire_buffers()
assert (wire is None) == bool(os.environ.get("RSX_NO_WIRE_PATH"))
rng = np.random.default_rng(4)
h = hashlib.sha256()
for t in range(24):
    cm = rng.uniform(-1, 1, (B, N, sim.cmd_dim)) * (30.0 if kind == 0 else 1.5)
    if kind == 1: cm[:, :, 0] = 0; cm[:, :, 4:] = 0
    ref.cmds_tensor().copy_(torch.from_numpy(cm.transpose(1, 2, 0).reshape(N * sim.cmd_dim, B).astype(np.float32)))
    ref.step_dev()
    want = ref.get_state_full()
    if wire is not None and t % 3 == 0:          # the zero-pass form: fill the pinned buffer, step, read the pinned buffer
        wire[0][...] = cm
        sim.step_wire()
        got_full = wire[1].copy()
        got = got_full[:, :sim.state_dim]
    elif t % 3 == 1:
        got = sim.step_state(np.ascontiguousarray(cm), copy=(t % 2 == 0)); got_full = sim.get_state_full()   # copy=False: a view of the pinned buffer
    else:
        sim.step(cm); got = sim.get_state(); got_full = sim.get_state_full()
    assert np.array_equal(got_full, want), t
    assert np.array_equal(got, want[:, :sim.state_dim]), t
    h.update(got_full.tobytes())
assert np.abs(want[:, 5] - (-0.2)).max() > 1e-3   # things moved
print("HASH", h.hexdigest())
'''
    res = []
    for env in ({}, {"RSX_NO_WIRE_PATH": "1"}):
        r = subprocess.run([sys.executable, "-c", child] + [str(v) for v in (kind, ft, nb, ny, B)], env=dict(os.environ, **env),
                           capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-3000:]
        res.append([l for l in r.stdout.splitlines() if l.startswith("HASH")][-1])
    assert res[0] == res[1]
