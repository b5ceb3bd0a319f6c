"""The N>1 path on CPU: two gloo ranks shard a population of envs and all-reduce the metrics
vector.  The per-rank work is done by the CPU oracle here (test double for the GPU engine):
what is under test is rsoccer_amd.dist — sharding covers every env id exactly once, and the
all-reduced metrics equal those of the unsharded population (int64, exact)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _population_metrics(O, base, count, steps, seed, max_steps):
    tot = np.zeros(8, dtype=np.int64)
    for i in range(count):
        e = O.OracleEnv(0, 0, 3, 3, 25, "f32")
        e.task_attach(1, seed, base + i, max_steps)
        e.task_reset()
        for _ in range(steps):
            e.task_step(None)
        tot += e.task_out()["metrics"]
    return tot


def _worker(rank, world, port, total, steps, seed, max_steps, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import oracle as O
    from rsoccer_amd import dist as rdist
    r, w, _ = rdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    base, count = rdist.shard(total)
    local = _population_metrics(O, base, count, steps, seed, max_steps)
    summed = rdist.allreduce_metrics(local)
    np.save(os.path.join(out, f"rank{rank}.npy"), np.concatenate([[base, count], local, summed]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_covers_every_env_once():
    from rsoccer_amd.dist import shard
    for total in (1, 7, 8, 4096, 32768, 33):
        for world in (1, 2, 3, 8):
            spans = [shard(total, r, world) for r in range(world)]
            ids = [i for b, c in spans for i in range(b, b + c)]
            assert ids == list(range(total))
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert [shard(32768, r, 8) for r in range(8)] == [(r * 4096, 4096) for r in range(8)]  # BASELINE configs[4]
    with pytest.raises(ValueError):
        shard(8, 3, 2)


def test_two_rank_gloo_metrics_allreduce(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    total, steps, seed, max_steps = 13, 90, 5, 40
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, steps, seed, max_steps, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npy").astype(np.int64) for r in range(2))
    assert (r0[0], r0[1]) == (0, 7) and (r1[0], r1[1]) == (7, 6)
    assert np.array_equal(r0[10:], r1[10:])                      # both ranks hold the same sum
    assert np.array_equal(r0[10:], r0[2:10] + r1[2:10])
    whole = _population_metrics(oracle_mod, 0, total, steps, seed, max_steps)
    assert np.array_equal(r0[10:], whole)                        # == the unsharded population
    assert whole[0] == total * steps and whole[1] >= total * (steps // max_steps)


def test_allreduce_is_identity_without_a_process_group():
    from rsoccer_amd.dist import allreduce_metrics
    m = np.arange(8, dtype=np.int64)
    assert np.array_equal(allreduce_metrics(m), m)


def _collective_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from rsoccer_amd import dist as rdist
    env_before = {k: os.environ.get(k) for k in ("TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCH_NCCL_ENABLE_MONITORING")}
    # the caller's own world, made before the collective exists
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coll = rdist.MetricsCollective(rank, world, device=None, prefer="nccl", timeout_s=5.0)   # no GPU here: degrades
    assert coll.ctl is not dist.group.WORLD and coll.backend == "gloo" and "rccl failed" in coll.describe()
    t = torch.arange(8, dtype=torch.int64) * (rank + 1)
    coll.all_reduce(t)
    coll.barrier()
    coll.close()
    assert dist.is_initialized()                      # close() left the caller's world alone ...
    u = torch.ones(1)
    dist.all_reduce(u)                                # ... and it still works
    env_after = {k: os.environ.get(k) for k in env_before}
    np.save(os.path.join(out, f"coll{rank}.npy"), np.concatenate([t.numpy(), [int(u[0])], [int(env_before == env_after)]]))
    dist.destroy_process_group()


def test_metrics_collective_uses_its_own_groups(tmp_path):
    """MetricsCollective inside a process that already runs torch.distributed — it makes a dedicated gloo
    control group, degrades without RCCL after agreeing over that group, destroys only what it made and leaves the
    NCCL environment switches as it found them."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_collective_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        v = np.load(tmp_path / f"coll{r}.npy")
        assert np.array_equal(v[:8], np.arange(8) * 3) and v[8] == 2 and v[9] == 1


def test_metrics_collective_needs_a_port_for_several_ranks(monkeypatch):
    from rsoccer_amd import dist as rdist
    for k in ("MASTER_PORT", "MASTER_ADDR"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        rdist.MetricsCollective(0, 2, prefer="gloo")
    coll = rdist.MetricsCollective(0, 1, prefer="gloo")   # one rank: a free port is picked
    assert coll.backend == "gloo" and coll.describe() == "gloo"
    coll.close()
    import torch.distributed as dist
    assert not dist.is_initialized()
