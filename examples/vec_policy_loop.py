"""Policy in the loop on the GPU: 4096 fused VSS-v0 envs stepped by a small torch MLP, nothing crosses PCIe.

    python examples/vec_policy_loop.py [num_envs] [steps] [--graph [ITERS]] [--fused-policy]

The call pattern is the reference's training loop (rsoccer_gym/vss/vss_gym_base.py:72-90, README.md:116-133): one
env.step(action) per policy action.  Every env.step() is one kernel launch of the engine; observations, rewards and flags
are torch views of the engine's buffers (no copies), the actions tensor is read in place.

Eager, that loop is bound by the launch overheads of the policy's small torch ops, not by the engine.  `--graph` captures
policy(obs) -> env.step(actions) (ITERS iterations per graph, default 1) into one hipGraph and replays it:
env.enable_graph_capture() moves the engine's step counter (the key of its per-step random draws) to device memory, so
a replayed step advances it exactly like an eager one — the run is bit-identical either way (tests/test_gpu_graph.py).
`--fused-policy` runs the same MLP as one hand-written kernel (examples/fused_policy.hip) instead of four library kernels.
(What the reference offers instead is one Python env object per process and ~5 k steps/s per core.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rsoccer_amd.vec import VecVSSEnv


def make_policy(obs_dim, act_dim, device, hidden=64, seed=0):
    """obs -> actions, a 2-layer tanh MLP written into `out` (four kernels: two addmm, two tanh; no copy)"""
    g = torch.Generator().manual_seed(seed)
    w1 = (torch.randn(obs_dim, hidden, generator=g) / obs_dim ** 0.5).to(device)
    b1 = torch.zeros(hidden, device=device)
    w2 = (torch.randn(hidden, act_dim, generator=g) / hidden ** 0.5).to(device)
    b2 = torch.zeros(act_dim, device=device)

    def policy(obs, out):
        h = torch.addmm(b1, obs, w1).tanh_()
        return torch.tanh(torch.addmm(b2, h, w2), out=out)
    return policy


def fused_policy_lib():
    """examples/fused_policy.hip built into examples/_build/libfused_policy.so (hipcc, gfx950; rebuilt when the source is newer)"""
    import ctypes
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src, out = os.path.join(here, "fused_policy.hip"), os.path.join(here, "_build", "libfused_policy.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call([os.environ.get("HIPCC", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", out, src])
    lib = ctypes.CDLL(out)
    vp = ctypes.c_void_p
    lib.fused_mlp_policy.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, vp]
    lib.fused_mlp_policy.restype = ctypes.c_int
    return lib


def make_fused_policy(obs_dim, act_dim, device, hidden=64, seed=0):
    """the same MLP (same weights as make_policy) as ONE hand-written kernel launch (examples/fused_policy.hip) on torch's current
    stream — capturable into a hipGraph like any launch.  What the policy-in-the-loop figure is when the policy is not four small
    library kernels: the engine's step is then most of the iteration."""
    assert hidden == 64
    g = torch.Generator().manual_seed(seed)
    w1 = (torch.randn(obs_dim, hidden, generator=g) / obs_dim ** 0.5).to(device).contiguous()
    b1 = torch.zeros(hidden, device=device)
    w2 = (torch.randn(hidden, act_dim, generator=g) / hidden ** 0.5).to(device).contiguous()
    b2 = torch.zeros(act_dim, device=device)
    lib = fused_policy_lib()
    keep = (w1, b1, w2, b2)

    def policy(obs, out):
        rc = lib.fused_mlp_policy(obs.data_ptr(), obs.shape[0], obs_dim, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                  out.data_ptr(), act_dim, torch.cuda.current_stream(device).cuda_stream)
        if rc:
            raise RuntimeError(f"fused_mlp_policy: {rc}")
        return out
    policy.keep = keep
    return policy


def run_eager(env, policy, actions, steps):
    obs = env._t["obs"]
    for _ in range(steps):
        env.step(policy(obs, actions))


def build_graph(env, policy, actions, iters=1, ret=None):
    """one hipGraph of `iters` x (policy -> env.step); `ret` (optional [B] tensor) accumulates the rewards inside the graph"""
    obs, reward = env._t["obs"], env._t["reward"]
    env.enable_graph_capture()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):          # torch's warm-up convention (real steps)
        env.step(policy(obs, actions))
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            env.step(policy(obs, actions))
            if ret is not None:
                ret += reward
    return g


def main():
    argv = [a for a in sys.argv[1:]]
    graph_iters = 0
    if "--graph" in argv:
        i = argv.index("--graph")
        graph_iters = int(argv[i + 1]) if i + 1 < len(argv) and argv[i + 1].isdigit() else 1
        del argv[i:i + (2 if i + 1 < len(argv) and argv[i + 1].isdigit() else 1)]
    fused = "--fused-policy" in argv
    if fused:
        argv.remove("--fused-policy")
    num_envs = int(argv[0]) if len(argv) > 0 else 4096
    steps = int(argv[1]) if len(argv) > 1 else 3000
    env = VecVSSEnv(num_envs, device=0, seed=0)
    policy = make_policy(env.sim.obs_dim, env.sim.act_dim, env.device)
    actions = torch.zeros(num_envs, env.sim.act_dim, device=env.device)
    if fused:   # one launch instead of four; checked against the torch form on the first observations
        env.reset()
        want = policy(env._t["obs"], torch.empty_like(actions)).clone()
        policy = make_fused_policy(env.sim.obs_dim, env.sim.act_dim, env.device)
        got = policy(env._t["obs"], torch.empty_like(actions))
        torch.cuda.synchronize()
        assert torch.allclose(got, want, atol=2e-5, rtol=0), float((got - want).abs().max())
    ret = torch.zeros(num_envs, device=env.device)
    env.reset()
    with torch.no_grad():
        run_eager(env, policy, actions, 100)          # warm-up
        if graph_iters:
            g = build_graph(env, policy, actions, graph_iters, ret)
            replays = max(1, steps // graph_iters)
            steps = replays * graph_iters
            for _ in range(5):
                g.replay()
            ret.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(replays):
                g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            how = f"hipGraph replay, {graph_iters} x (policy -> step) per graph"
        else:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            obs, reward = env._t["obs"], env._t["reward"]
            for _ in range(steps):
                env.step(policy(obs, actions))
                ret += reward
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            how = "eager"
    m = env.metrics()
    print(f"{num_envs} envs x {steps} steps with a 40-64-2 MLP policy{' as one fused kernel' if fused else ''} ({how}): {num_envs * steps / dt / 1e6:.1f} M env-steps/s "
          f"({dt / steps * 1e6:.1f} us per vector step)")
    print(f"episodes finished {m['episodes']}, goals for / against {m['goals_for']} / {m['goals_against']}, "
          f"mean reward per step {float(ret.mean()) / steps:+.4f}")
    blob = env.checkpoint()                       # a run can be stopped here and continued elsewhere
    print(f"checkpoint: {blob.size / 1e6:.2f} MB")
    env.close()


if __name__ == "__main__":
    main()
