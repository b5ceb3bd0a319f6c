"""Metrics all-reduce on the compute stream vs on a side stream behind a timing-less event
(development tool; one rank, nccl)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from rsoccer_amd import _lib as L
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sim = L.Sim(0, 0, 3, 3, 25, 4096); sim.task_attach(1, 0, 0, 0); sim.task_reset()
tens = sim.task_tensors()
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
mbuf = torch.zeros(8, dtype=torch.int64, device="cuda")
ev = torch.cuda.Event(enable_timing=False)
def none(): pass
def inline():
    sim.metrics_fold(torch.cuda.current_stream().cuda_stream); mbuf.copy_(tens["metrics"], non_blocking=True); dist.all_reduce(mbuf)
def sidestream():
    ev.record(main); side.wait_event(ev)
    with torch.cuda.stream(side):
        sim.metrics_fold(torch.cuda.current_stream().cuda_stream); mbuf.copy_(tens["metrics"], non_blocking=True); dist.all_reduce(mbuf)
for name, fn in (("no all-reduce", none), ("on the compute stream", inline), ("side stream + event", sidestream), ("no all-reduce", none)):
    for _ in range(5):
        sim.task_step_n(100, main.cuda_stream); fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        sim.task_step_n(100, main.cuda_stream); fn()
    torch.cuda.synchronize()
    print(f"{name:26s} {(time.perf_counter() - t) / 1e4 * 1e6:6.3f} us/step")
dist.destroy_process_group()
