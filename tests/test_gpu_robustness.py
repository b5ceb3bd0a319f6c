"""What include/rsx.h promises about handles beyond single-threaded use at benchmark sizes: independent handles on independent
threads and streams, arrays between 2 and 4 GB (the large-batch kernels address rows with 32-bit byte offsets below 2 GB and hand
larger handles to the lane-group kernels), and no leak over many create / destroy cycles."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _checksum(torch, t):
    """order-sensitive 64-bit checksum of a float32 tensor's bit patterns (on the device)"""
    b = t.contiguous().view(torch.int32).to(torch.int64).flatten() & 0xFFFFFFFF
    idx = torch.arange(b.numel(), device=b.device, dtype=torch.int64)
    return int(((b * ((idx % 1000003) + 1)) % 2305843009213693951).sum().item() % 2305843009213693951)


def _run(torch, L, cfg, steps, stream):
    kind, ft, nb, ny, task, B, seed = cfg
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(task, seed, 0, 0)
    s = stream.cuda_stream
    sim.task_reset(s)
    for _ in range(steps // 50):
        sim.task_step_n(25, s)
        for _ in range(25):
            sim.task_step(None, s)
    stream.synchronize()
    t = sim.task_tensors()
    out = (_checksum(torch, sim.state_tensor()), _checksum(torch, t["obs"]), _checksum(torch, t["reward"]),
           sim.read_metrics(s).tolist())
    sim.close()
    return out


def test_two_handles_on_two_threads_and_streams_are_independent():
    """rsx.h: "a handle is not thread-safe; distinct handles are independent".  VSS-v0 at 4096 envs and SSLStaticDefenders at 2048,
    2000 steps each, stepped concurrently from two Python threads on two streams: both bit-identical to their solo runs."""
    import torch
    from rsoccer_amd import _lib as L
    cfgs = [(0, 0, 3, 3, 1, 4096, 11), (1, 2, 1, 6, 2, 2048, 12)]
    solo = [_run(torch, L, c, 2000, torch.cuda.Stream()) for c in cfgs]
    res, err = [None, None], []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            res[i] = _run(torch, L, cfgs[i], 2000, torch.cuda.Stream())
        except Exception as ex:   # surfaced below: an exception in a thread must fail the test
            err.append(repr(ex))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    assert res[0] == solo[0] and res[1] == solo[1]
    assert solo[0][3][0] == 4096 * 2000 and solo[1][3][0] == 2048 * 2000


def test_a_handle_between_2_and_4_gb_steps_like_its_shards():
    """SSLStaticDefenders 1v6 at 7 M envs: 2.35 GB of state — beyond the 32-bit byte offsets of the one-lane-per-env kernel, so
    the library keeps the lane-group kernels (rsx_task_layout says so); 20 steps give the same state, observation and reward
    bits as seven 1 M-env handles with the matching env_id_base."""
    import torch
    from rsoccer_amd import _lib as L
    free, _ = torch.cuda.mem_get_info()
    B, shards, seed, steps = 7 * (1 << 20), 7, 5, 20
    if free < 16 << 30:
        pytest.skip(f"only {free >> 30} GiB free")
    s = torch.cuda.current_stream().cuda_stream
    big = L.Sim(1, 2, 1, 6, 25, B)
    big.task_attach(2, seed, 0, 0)
    assert big.task_layout() == "8-lanes-per-env"
    assert (big.state_dim + 2) * B * 4 >= 1 << 31
    big.task_reset(s)
    big.task_step_n(steps, s)
    torch.cuda.synchronize()
    bt = big.task_tensors()
    state, obs, rew = big.state_tensor(), bt["obs"], bt["reward"]
    n = B // shards
    for k in range(shards):
        sh = L.Sim(1, 2, 1, 6, 25, n)
        sh.task_attach(2, seed, k * n, 0)
        assert sh.task_layout() == "one-lane-per-env"
        sh.task_reset(s)
        sh.task_step_n(steps, s)
        torch.cuda.synchronize()
        t = sh.task_tensors()
        assert torch.equal(sh.state_tensor().view(torch.int32), state[:, k * n:(k + 1) * n].view(torch.int32)), f"state of shard {k}"
        assert torch.equal(t["obs"].view(torch.int32), obs[k * n:(k + 1) * n].view(torch.int32)), f"obs of shard {k}"
        assert torch.equal(t["reward"].view(torch.int32), rew[k * n:(k + 1) * n].view(torch.int32)), f"reward of shard {k}"
        sh.close()
        del sh, t
    m = big.read_metrics()
    assert m[0] == B * steps
    big.close()


def test_create_destroy_cycles_leave_device_memory_unchanged():
    """1000 x (rsx_create, rsx_task_attach, a reset and a step, rsx_destroy): hipMemGetInfo afterwards is what it was."""
    import torch
    from rsoccer_amd import _lib as L
    s = torch.cuda.current_stream().cuda_stream

    def cycle(k):
        kind, ft, nb, ny, task = [(0, 0, 3, 3, 1), (1, 2, 1, 6, 2), (1, 1, 11, 11, 7)][k % 3]
        sim = L.Sim(kind, ft, nb, ny, 25, 256 + 64 * (k % 5))
        sim.task_attach(task, k, 0, 0)
        sim.task_reset(s)
        sim.task_step(None, s)
        if k % 7 == 0:
            sim.task_enable_capture(s)
            sim.task_step(None, s)
        if k % 11 == 0:
            sim.state_buffers()           # allocates the second state buffer
            for _ in range(1 + k % 2):    # ... whose role the flips trade with the first one's: an odd number of flips leaves the
                sim.step_dev_flip(s)      # handle's CURRENT buffer in the second allocation (rsx_destroy frees the allocations, not the roles)
            torch.cuda.synchronize()
        sim.close()
        assert L.drop_pending_hip_error() == 0, k     # nothing this library did left an error in the thread's HIP slot

    for k in range(20):                   # warm the allocator's pools
        cycle(k)
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info()[0]
    for k in range(1000):
        cycle(k)
    torch.cuda.synchronize()
    after = torch.cuda.mem_get_info()[0]
    assert abs(after - before) <= 8 << 20, (before, after)   # (the driver hands memory back in 2 MB pieces)
