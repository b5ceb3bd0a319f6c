"""Counter-calibration target (tools/prof_calibration.sh runs it under rocprofv3 --pmc): launches that move a KNOWN number of
bytes in the step kernels' own access patterns — rsx_task_rollout(0): every row a multi-step launch loads is loaded, every row it
stores is stored, no step in between.  Rows per env (4 bytes each), counted from the kernels' load / store code:

  vss  one-lane-per-env   (rsx_epl.hpp, MODE_ROLLOUT):      reads 36 robot + 10 OU + 7 ball + steps, episode, info 1-3 = 58; writes 36 + 10 OU
                                                             + ball x, y, vx, vy + steps + episode = 52 (height / vz / spin rows only when off rest:
                                                             never right after a reset)
  vss  8-lanes-per-env    (rsx_kernels.hpp, MODE_ROLLOUT):   reads 36 + 7 ball + steps, episode + 10 OU + info 1-3 + prev_pot = 59; writes 36 + 7 ball
                                                             + steps, episode + 10 OU + prev_pot = 56
  sd   one-lane-per-env   (rsx_epl_ssl.hpp, MODE_ROLLOUT):  reads 42 robot + 7 ball + steps, episode, info 0-7, ep_ret = 60; writes 77 robot (pose 6 +
                                                             infrared + 4 wheels) + 7 ball + steps, episode, ep_ret = 87
  python tools/calib_target.py <vss|sd>:<envs>[:lanes] [launches]"""
import os
import sys
if sys.argv[1].endswith(":lanes"):
    os.environ["RSX_LAYOUT"] = "lanes"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L

TASKS = {"vss": (0, 0, 3, 3, 1), "sd": (1, 2, 1, 6, 2)}
ROWS = {("vss", "one-lane-per-env"): (58, 52), ("vss", "8-lanes-per-env"): (59, 56), ("sd", "one-lane-per-env"): (60, 87)}
parts = sys.argv[1].split(":")
name, envs = parts[0], int(parts[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kind, ft, nb, ny, task = TASKS[name]
sim = L.Sim(kind, ft, nb, ny, 25, envs)
sim.task_attach(task, 0, 0, 0)
sim.task_reset()
torch.cuda.synchronize()
for _ in range(n):
    sim.task_rollout(0)
torch.cuda.synchronize()
lay = sim.task_layout()
r, w = ROWS[(name, lay)]
print("calib", sys.argv[1], "layout", lay, "launches", n, "read_bytes", r * 4 * envs, "write_bytes", w * 4 * envs, flush=True)
sim.close()
