"""Profiling target: a VSS-v0-shaped task written with the BATCHED HOOKS (rsoccer_amd.vec.VecVSSBaseEnv),
4096 envs, device placements, TimeLimit + same-step auto-reset: 300 steps.  Run under
`rocprofv3 --memory-copy-trace --stats` to show that step() moves nothing between host and device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd.vec import VecVSSBaseEnv


class Task(VecVSSBaseEnv):
    def __init__(self, n):
        super().__init__(0, 3, 3, 0.025, n, max_episode_steps=60)
        self.gen = torch.Generator(device="cuda"); self.gen.manual_seed(1)

    def _get_commands(self, action):
        v = torch.clamp(action * self.max_v, -self.max_v, self.max_v) / self.field.rbt_wheel_radius
        self.commands[0, 0].copy_(v[:, 0]); self.commands[0, 1].copy_(v[:, 1])

    def _frame_to_observations(self):
        f = self.frame
        cols = [self.norm_pos(f.ball.x), self.norm_pos(f.ball.y), self.norm_v(f.ball.v_x), self.norm_v(f.ball.v_y)]
        for team in (f.robots_blue, f.robots_yellow):
            for i in range(3):
                cols += [self.norm_pos(team[i].x), self.norm_pos(team[i].y), self.norm_v(team[i].v_x), self.norm_v(team[i].v_y)]
        return torch.stack(cols, 1)

    def _calculate_reward_and_done(self):
        return self.frame.ball.x - self.last_frame.ball.x, self.frame.ball.x.abs() > self.field.length / 2

    def _get_initial_positions(self):
        B = self.num_envs
        ball = torch.zeros(B, 4, device="cuda"); ball[:, :2] = torch.rand(B, 2, device="cuda", generator=self.gen) * 0.6 - 0.3
        blue = torch.zeros(B, 3, 3, device="cuda"); yellow = torch.zeros(B, 3, 3, device="cuda")
        for k in range(3):
            blue[:, k, 0] = -0.5; blue[:, k, 1] = 0.3 * (k - 1)
            yellow[:, k, 0] = 0.5; yellow[:, k, 1] = 0.3 * (k - 1); yellow[:, k, 2] = 180.0
        return ball, blue, yellow


env = Task(4096)
obs, _ = env.reset()
act = torch.rand(4096, 2, device="cuda") * 2 - 1
torch.cuda.synchronize()
print("STEPS BEGIN", flush=True)
for _ in range(300):
    obs, rew, done, trunc, info = env.step(act)
torch.cuda.synchronize()
print("STEPS END", float(rew.sum()), int(env.steps.max()), flush=True)
env.close()
