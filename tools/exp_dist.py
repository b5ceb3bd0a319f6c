import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.zeros(8, dtype=torch.int64, device="cuda")
for name, fn in (("barrier", lambda: dist.barrier()), ("all_reduce sync", lambda: dist.all_reduce(x)),
                 ("all_reduce async+wait", lambda: dist.all_reduce(x, async_op=True).wait())):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(name, (time.perf_counter() - t) / 20 * 1e6, "us")
dist.destroy_process_group()
