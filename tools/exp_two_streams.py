"""Development: do independent halves of the batch on separate streams overlap their launch tails?
Compares 1 handle x 4096 envs on one stream with k handles x 4096/k envs on k streams (env-steps/s)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L
TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for k in (1, 2, 4):
    B = TOTAL // k
    sims, streams = [], []
    for i in range(k):
        s = L.Sim(0, 0, 3, 3, 25, B); s.task_attach(1, 0, i * B, 0); s.task_reset()
        sims.append(s); streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    def run(n, chunk=50):
        done = 0
        while done < n:
            for s, st in zip(sims, streams):
                s.task_step_n(chunk, st.cuda_stream)
            done += chunk
    run(1000); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(4000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{k} stream(s) x {B} envs: {dt / 4000 * 1e6:.2f} us per step of all {TOTAL} envs = {TOTAL * 4000 / dt:.3e} env-steps/s", flush=True)
    for s in sims: s.close()
