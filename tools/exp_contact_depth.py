"""How deep are the partner walks of the headline population?  CPU oracle, VSS-v0, 4096 envs in steady state, contact sets from the
states at the end of a step: partners per body, touching pairs per env and per wave of eight envs (LABBOOK round 6)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
O.build()
B = 4096
envs = [O.OracleEnv(0, 0, 3, 3, 25, "f32") for _ in range(B)]
for e, r in enumerate(envs):
    r.task_attach(1, 0, e, 0); r.task_reset()
O.set_threads(16)
O.vec_task_step(envs, 400, "f32")
fp = envs[0].field_params()
r_rob, r_ball = fp[14], fp[6]
rs_rr, rs_rb = 2 * r_rob, r_rob + r_ball
hist_D = np.zeros(8, int); hist_P = np.zeros(22, int)
waveD = []; waveP = []
for it in range(30):
    O.vec_task_step(envs, 7, "f32")
    st = np.stack([r.get_state() for r in envs])
    pos = np.zeros((B, 7, 2))
    pos[:, 6] = st[:, 0:2]
    for k in range(6):
        pos[:, k] = st[:, 5 + 6 * k: 7 + 6 * k]
    d = np.linalg.norm(pos[:, :, None] - pos[:, None, :], axis=-1)
    thr = np.full((7, 7), rs_rr); thr[6, :] = rs_rb; thr[:, 6] = rs_rb
    touch = (d < thr) & (d > 0)
    D = touch.sum(-1)              # partners per body [B,7]
    P = touch.sum((1, 2)) // 2     # pairs per env
    for v in D.max(1): hist_D[v] += 1
    for v in P: hist_P[v] += 1
    wD = D.max(1).reshape(-1, 8).max(1)       # per wave (8 envs): deepest lane
    wP = P.reshape(-1, 8).sum(1)
    waveD.append(wD); waveP.append(wP)
waveD = np.array(waveD); waveP = np.array(waveP)
print("per env: max partners of a body", hist_D / hist_D.sum())
print("per env: touching pairs", (hist_P / hist_P.sum())[:8])
print("per wave (8 envs): deepest lane D: ", np.bincount(waveD.ravel(), minlength=6) / waveD.size)
print("per wave: pairs P mean %.2f, p95 %d, max %d" % (waveP.mean(), np.percentile(waveP, 95), waveP.max()))
print("per launch (512 waves): max D over waves:", np.bincount(waveD.max(1)), " max P:", waveP.max(1).mean())
# modelled contact cycles per sweep: now c*D ; pair-parallel: c2 (one pair, both sides) * ceil(P/8 per env...) ~ c2 + tail*Dmax
c, ov = 385, 300
now = ov + c * waveD
new = np.where(waveP > 0, ov + 250 + 520 + 40 * waveD, ov)   # scan/assign + one pair evaluation (both sides) + serial FMA tail per partner
print("mean wave contact cycles/sweep now %.0f new %.0f ; slowest wave of a launch now %.0f new %.0f" % (now.mean(), new.mean(), now.max(1).mean(), new.max(1).mean()))
