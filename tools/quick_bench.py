"""Quick throughput probe of the fused VSS-v0 step (development tool; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L

def run(B, kind=0, ft=0, nb=3, ny=3, task=1, K=2000):
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(task, 0, 0, 0)
    sim.task_reset()
    torch.cuda.synchronize()
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    for name, fn in (("step", lambda n: [sim.task_step(None, s) for _ in range(n)]),
                     ("graph", lambda n: sim.task_step_n(n, s)),
                     ("rollout", lambda n: sim.task_rollout(n, s))):
        fn(200); torch.cuda.synchronize()
        t = time.perf_counter(); fn(K); torch.cuda.synchronize(); dt = time.perf_counter() - t
        out[name] = (dt / K * 1e6, B * K / dt)
    m = sim.read_metrics()
    sim.close()
    print(f"B={B:8d} kind={kind} N={nb+ny} " + "  ".join(f"{k}: {v[0]:8.2f} us/step {v[1]/1e6:9.1f} M env-steps/s" for k, v in out.items()), "episodes", m[1], flush=True)

if __name__ == "__main__" and len(sys.argv) > 1:
    for a in sys.argv[1:]:
        run(int(a), K=2000 if int(a) <= 32768 else 200)
    sys.exit(0)

if __name__ == "__main__":
    for B in (4096, 32768, 262144, 1048576):
        run(B, K=2000 if B <= 32768 else 200)
    run(2048, 1, 2, 1, 6, 2)
    run(65536, 1, 2, 1, 6, 2, K=200)
