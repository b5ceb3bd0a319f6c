import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsoccer_amd import _lib as L
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sim = L.Sim(0, 0, 3, 3, 25, 4096); sim.task_attach(1, 0, 0, 0); sim.task_reset()
tens = sim.task_tensors()
main = torch.cuda.current_stream(); s = main.cuda_stream
side = torch.cuda.Stream()
mbuf = torch.zeros(8, dtype=torch.int64, device="cuda")
pend = []
def v_none(): pass
def v_copy_side():
    ev = torch.cuda.Event(); ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev); sim.metrics_fold(torch.cuda.current_stream().cuda_stream); mbuf.copy_(tens["metrics"], non_blocking=True)
def v_ar_side():
    ev = torch.cuda.Event(); ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev); pend.append(dist.all_reduce(mbuf, async_op=True))
def v_ar_main(): pend.append(dist.all_reduce(mbuf, async_op=True))
def v_ar_main_sync(): dist.all_reduce(mbuf)
def v_event_only():
    ev = torch.cuda.Event(); ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev)
allv = {"none": v_none, "event": v_event_only, "copyside": v_copy_side, "arside": v_ar_side, "armain": v_ar_main, "armainsync": v_ar_main_sync}
for name, fn in [(k, allv[k]) for k in sys.argv[1:]]:
    sim.task_step_n(400, s); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(15):
        sim.task_step_n(200, s); fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    for p in pend: p.wait()
    del pend[:]
    print(f"{name:28s} {dt / 3000 * 1e6:7.2f} us/step")
dist.destroy_process_group()
