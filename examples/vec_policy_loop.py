"""Policy in the loop on the GPU: 4096 fused VSS-v0 envs stepped by a small torch MLP, nothing crosses PCIe.

    python examples/vec_policy_loop.py [num_envs] [steps]

Every env.step() is one kernel launch of the engine; observations, rewards and flags are torch views of the engine's
buffers (no copies), the actions tensor is read in place.  Prints env-steps/s and what the episodes looked like.
(What the reference offers instead is one Python env object per process and ~5 k steps/s per core.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rsoccer_amd.vec import VecVSSEnv


def main():
    num_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    env = VecVSSEnv(num_envs, device=0, seed=0)
    policy = torch.nn.Sequential(torch.nn.Linear(env.sim.obs_dim, 64), torch.nn.Tanh(), torch.nn.Linear(64, env.sim.act_dim),
                                 torch.nn.Tanh()).to(env.device)
    obs, _ = env.reset()
    ret = torch.zeros(num_envs, device=env.device)
    with torch.no_grad():
        for _ in range(100):                      # warm-up
            obs, reward, terminated, truncated, info = env.step(policy(obs))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            obs, reward, terminated, truncated, info = env.step(policy(obs))
            ret += reward
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    m = env.metrics()
    print(f"{num_envs} envs x {steps} steps with a 40-64-2 MLP policy: {num_envs * steps / dt / 1e6:.1f} M env-steps/s "
          f"({dt / steps * 1e6:.1f} us per vector step)")
    print(f"episodes finished {m['episodes']}, goals for / against {m['goals_for']} / {m['goals_against']}, "
          f"mean reward per step {float(ret.mean()) / steps:+.4f}")
    blob = env.checkpoint()                       # a run can be stopped here and continued elsewhere
    print(f"checkpoint: {blob.size / 1e6:.2f} MB")
    env.close()


if __name__ == "__main__":
    main()
