"""Pieces shared by the SSL hardware-challenge tasks: the global->local velocity conversion
(identical in static_defenders.py:132-148, dribbling.py:119-135, contested_possession.py:121-137)
and the two distance-progress terms of the shaped rewards."""
import numpy as np


def convert_actions(action, angle, max_v, max_w):
    """de-normalise, rotate the global velocity into the robot frame, cap its norm at max_v"""
    gx, gy = action[0] * max_v, action[1] * max_v
    v_theta = action[2] * max_w
    c, s = np.cos(angle), np.sin(angle)
    v_x, v_y = gx * c + gy * s, -gx * s + gy * c
    speed = np.linalg.norm([v_x, v_y])
    scale = 1 if speed < max_v else max_v / speed
    return v_x * scale, v_y * scale, v_theta


def _dist(ax, ay, bx, by):
    return np.linalg.norm(np.array([ax, ay]) - np.array([bx, by]))


def robot_ball_approach(last_frame, frame):
    """decrease of the blue-0 <-> ball distance over the step, clipped to +-1"""
    before = _dist(last_frame.robots_blue[0].x, last_frame.robots_blue[0].y, last_frame.ball.x, last_frame.ball.y)
    after = _dist(frame.robots_blue[0].x, frame.robots_blue[0].y, frame.ball.x, frame.ball.y)
    return np.clip(before - after, -1, 1)


def ball_progress_to(px, py, last_frame, frame):
    """decrease of the ball <-> (px, py) distance over the step, clipped to +-1"""
    before = _dist(px, py, last_frame.ball.x, last_frame.ball.y)
    after = _dist(px, py, frame.ball.x, frame.ball.y)
    return np.clip(before - after, -1, 1)


def blue_observation(env, robot, with_velocity=True):
    heading = np.deg2rad(robot.theta)
    out = [env.norm_pos(robot.x), env.norm_pos(robot.y), np.sin(heading), np.cos(heading)]
    if with_velocity:
        out += [env.norm_v(robot.v_x), env.norm_v(robot.v_y)]
    out.append(env.norm_w(robot.v_theta))
    return out


def observation_entries(n_blue, n_yellow, with_velocity=True, infrared="flag", lead=0):
    """``(state index, kind)`` per observation slot of the SSL hardware-challenge layouts (static_defenders.py:90-112,
    dribbling.py:76-104, contested_possession.py:78-104, pass_endurance.py:77-91) for ``VSSBaseEnv._observation_plan``:
    ``lead`` task-specific leading slots (filled by the task), ball x, y, v_x, v_y, then per blue robot x, y, sin, cos
    [, v_x, v_y], v_theta, infrared and per yellow robot x, y.  SSL robots are 11 values wide in the state vector."""
    entries = [(0, "const")] * lead + [(0, "pos"), (1, "pos"), (3, "v"), (4, "v")]
    for i in range(n_blue):
        b = 5 + 11 * i
        entries += [(b, "pos"), (b + 1, "pos"), (b + 2, "sin"), (b + 2, "cos")]
        if with_velocity:
            entries += [(b + 3, "v"), (b + 4, "v")]
        entries += [(b + 5, "w"), (b + 6, infrared)]
    for i in range(n_yellow):
        b = 5 + 11 * (n_blue + i)
        entries += [(b, "pos"), (b + 1, "pos")]
    return entries


def observe(env, lead_value=None, **layout):
    """the task's observation from the state vector behind ``env.frame`` (plan cached on the env, rebuilt when a task
    changes its speed limits); frames assembled by hand fall back to ``None`` (the caller then walks the records)"""
    state = env.frame.state
    if state is None:
        return None
    plan = env.__dict__.get("_obs_plan")
    if plan is None or plan["key"] != (env.max_pos, env.max_v, env.max_w):
        plan = env._obs_plan = env._observation_plan(observation_entries(env.n_robots_blue, env.n_robots_yellow, **layout))
    return env._observe(plan, state, lead_value)
