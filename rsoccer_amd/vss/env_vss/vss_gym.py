"""VSS-v0: one learning robot (blue 0) in a 3v3 IEEE VSS match, the other five robots driven by
Ornstein-Uhlenbeck noise.  Restates the task of rsoccer_gym/vss/env_vss/vss_gym.py:13-311 on top
of :class:`VSSBaseEnv`.

Observation Box(40), bounds +-1.2 (:93-117):
    0..3    ball x, y, v_x, v_y (normalised)
    4+7i    blue i: x, y, sin(theta), cos(theta), v_x, v_y, v_theta
    25+5i   yellow i: x, y, v_x, v_y, v_theta
Action Box(2): left / right wheel speed of blue 0 as a fraction of the maximum.
Reward: +-10 on a goal, otherwise 0.2 * move-to-ball + 0.8 * ball-potential gradient
+ 2e-4 * energy penalty (:144-192).  The episode ends on a goal; ``gym.make`` adds the 1200-step
TimeLimit.  The fused, batched version of this task is ``rsoccer_amd.vec.VecVSSEnv``.
"""
import math
import random
from typing import Dict

import numpy as np

from rsoccer_amd import gymshim as gym
from rsoccer_amd.Entities import Ball, Frame, Robot
from rsoccer_amd.Utils import KDTree, OrnsteinUhlenbeckAction
from rsoccer_amd.vss.vss_gym_base import VSSBaseEnv

_INFO_KEYS = ("goal_score", "move", "ball_grad", "energy", "goals_blue", "goals_yellow")
_W_MOVE, _W_BALL_GRAD, _W_ENERGY = 0.2, 0.8, 2e-4


class VSSEnv(VSSBaseEnv):
    def __init__(self, render_mode=None, sim_backend=None):
        super().__init__(field_type=0, n_robots_blue=3, n_robots_yellow=3, time_step=0.025,
                         render_mode=render_mode, sim_backend=sim_backend)
        self.action_space = gym.spaces.Box(low=-1, high=1, shape=(2,), dtype=np.float32)
        self.observation_space = gym.spaces.Box(low=-self.NORM_BOUNDS, high=self.NORM_BOUNDS,
                                                shape=(40,), dtype=np.float32)
        self.previous_ball_potential = None
        self.actions: Dict = None
        self.reward_shaping_total = None
        self.v_wheel_deadzone = 0.05
        n = self.n_robots_blue + self.n_robots_yellow
        self.ou_actions = [OrnsteinUhlenbeckAction(self.action_space, dt=self.time_step) for _ in range(n)]

    def reset(self, *, seed=None, options=None):
        self.actions = None
        self.reward_shaping_total = None
        self.previous_ball_potential = None
        for ou in self.ou_actions:
            ou.reset()
        return super().reset(seed=seed, options=options)

    def step(self, action):
        observation, reward, terminated, truncated, _ = super().step(action)
        return observation, reward, terminated, truncated, self.reward_shaping_total

    # ---- hooks ----
    def _frame_to_observations(self):
        f = self.frame
        obs = [self.norm_pos(f.ball.x), self.norm_pos(f.ball.y), self.norm_v(f.ball.v_x), self.norm_v(f.ball.v_y)]
        for i in range(self.n_robots_blue):
            r = f.robots_blue[i]
            heading = np.deg2rad(r.theta)
            obs += [self.norm_pos(r.x), self.norm_pos(r.y), np.sin(heading), np.cos(heading),
                    self.norm_v(r.v_x), self.norm_v(r.v_y), self.norm_w(r.v_theta)]
        for i in range(self.n_robots_yellow):
            r = f.robots_yellow[i]
            obs += [self.norm_pos(r.x), self.norm_pos(r.y), self.norm_v(r.v_x), self.norm_v(r.v_y),
                    self.norm_w(r.v_theta)]
        return np.array(obs, dtype=np.float32)

    def _get_commands(self, actions):
        self.actions = {0: actions}
        commands = [self._wheel_command(False, 0, actions)]
        # the other robots follow OU noise; note ou_actions[0] is never sampled (vss_gym.py:127-140)
        for i in range(1, self.n_robots_blue):
            noise = self.ou_actions[i].sample()
            self.actions[i] = noise
            commands.append(self._wheel_command(False, i, noise))
        for i in range(self.n_robots_yellow):
            commands.append(self._wheel_command(True, i, self.ou_actions[self.n_robots_blue + i].sample()))
        return commands

    def _wheel_command(self, yellow, idx, action):
        left, right = self._actions_to_v_wheels(action)
        return Robot(yellow=yellow, id=idx, v_wheel0=left, v_wheel1=right)

    def _actions_to_v_wheels(self, actions):
        """fraction of max speed -> wheel rad/s, with saturation and a 0.05 m/s dead zone"""
        speeds = np.clip((actions[0] * self.max_v, actions[1] * self.max_v), -self.max_v, self.max_v)
        out = []
        for v in speeds:
            if -self.v_wheel_deadzone < v < self.v_wheel_deadzone:
                v = 0
            out.append(v / self.field.rbt_wheel_radius)
        return out[0], out[1]

    def _calculate_reward_and_done(self):
        if self.reward_shaping_total is None:
            self.reward_shaping_total = dict.fromkeys(_INFO_KEYS, 0)
        total = self.reward_shaping_total
        half_length = self.field.length / 2
        if self.frame.ball.x > half_length:
            total["goal_score"] += 1
            total["goals_blue"] += 1
            return 10, True
        if self.frame.ball.x < -half_length:
            total["goal_score"] -= 1
            total["goals_yellow"] += 1
            return -10, True
        if self.last_frame is None:
            return 0, False
        grad = self._ball_grad()
        move = self._move_reward()
        energy = self._energy_penalty()
        total["move"] += _W_MOVE * move
        total["ball_grad"] += _W_BALL_GRAD * grad
        total["energy"] += _W_ENERGY * energy
        return _W_MOVE * move + _W_BALL_GRAD * grad + _W_ENERGY * energy, False

    def _get_initial_positions_frame(self):
        """ball and robots uniformly on the field, at least 0.1 m apart (vss_gym.py:194-233);
        draws come from python's global ``random`` in the reference's order"""
        half_length, half_width = self.field.length / 2, self.field.width / 2
        rand_x = lambda: random.uniform(-half_length + 0.1, half_length - 0.1)
        rand_y = lambda: random.uniform(-half_width + 0.1, half_width - 0.1)
        frame = Frame()
        frame.ball = Ball(x=rand_x(), y=rand_y())
        placed = KDTree()
        placed.insert((frame.ball.x, frame.ball.y))
        for team, count in ((frame.robots_blue, self.n_robots_blue), (frame.robots_yellow, self.n_robots_yellow)):
            for i in range(count):
                pos = (rand_x(), rand_y())
                while placed.get_nearest(pos)[1] < 0.1:
                    pos = (rand_x(), rand_y())
                placed.insert(pos)
                team[i] = Robot(x=pos[0], y=pos[1], theta=random.uniform(0, 360))
        return frame

    # ---- reward terms ----
    def _ball_grad(self):
        """change of the ball 'potential' (closer to the attacked goal = higher) per second"""
        length_cm = self.field.length * 100
        half_len = (self.field.length / 2.0) + self.field.goal_depth
        dx_defence = (half_len + self.frame.ball.x) * 100
        dx_attack = (half_len - self.frame.ball.x) * 100
        dy = self.frame.ball.y * 100
        potential = ((-math.sqrt(dx_attack ** 2 + 2 * dy ** 2) + math.sqrt(dx_defence ** 2 + 2 * dy ** 2)) / length_cm - 1) / 2
        grad = 0
        if self.previous_ball_potential is not None:
            grad = np.clip((potential - self.previous_ball_potential) * 3 / self.time_step, -5.0, 5.0)
        self.previous_ball_potential = potential
        return grad

    def _move_reward(self):
        """speed of blue 0 along the direction to the ball, scaled by 0.4 m/s"""
        r = self.frame.robots_blue[0]
        to_ball = np.array([self.frame.ball.x, self.frame.ball.y]) - np.array([r.x, r.y])
        to_ball = to_ball / np.linalg.norm(to_ball)
        return np.clip(np.dot(to_ball, np.array([r.v_x, r.v_y])) / 0.4, -5.0, 5.0)

    def _energy_penalty(self):
        cmd = self.sent_commands[0]
        return -(abs(cmd.v_wheel0) + abs(cmd.v_wheel1))
