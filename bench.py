#!/usr/bin/env python
"""Headline benchmark: env-steps/s of the fused VSS-v0 3v3 step at 4096 envs per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode step|rollout] [--envs B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batched env.step() of every env on the rank (BASELINE.json configs[1]:
"VSS-v0 3v3, 4096 batched envs on 1xMI355X, random actions"): device-side random agent action,
OU noise for the other five robots, 5 physics sub-steps, observation, reward, done, TimeLimit
and same-step auto-reset, in ONE kernel launch per step (mode `step`, the default and the value
reported).  All inputs are resident in HBM before the timed region.  With N > 1 each rank owns
envs [rank*B, (rank+1)*B) (weak scaling, no data-path collective); a 64-byte metrics vector is
all-reduced over RCCL every 100 steps (a ~20 us stream-ordered collective).

Prints ONE JSON line on rank 0 (see the driver contract); extra keys: `roofline`,
`cpu_baseline`, `rollout` (the same work with all K steps inside one launch).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 541   # VSS-v0 fused = 2*164 (state r/w) + 48 (cmds) + 160 (obs) + 4 + 1
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def usable_cores():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def ensure_built(local_rank):
    """The HIP library normally travels prebuilt (__graft_entry__.build()); on a checkout without
    it, local rank 0 compiles it (atomic rename) and the other ranks wait for the file."""
    so = os.path.join(ROOT, "rsoccer_amd", "librsx_hip.so")
    if os.path.exists(so):
        return
    if local_rank == 0:
        import __graft_entry__ as g
        g.build()
        return
    t0 = time.time()
    while not os.path.exists(so):
        if time.time() - t0 > 900:
            raise SystemExit(f"{so} did not appear (local rank 0 builds it)")
        time.sleep(0.5)


def cpu_baseline(envs, budget_s=12.0, single_s=1.5):
    """Times the CPU oracle (oracle/, float instantiation, OpenMP over envs) on a bounded sample
    of the same workload.  This is the only place bench.py touches oracle/."""
    from oracle import oracle as O
    O.build()
    cores = usable_cores()
    O.set_threads(cores)
    es = []
    for i in range(envs):
        e = O.OracleEnv(0, 0, 3, 3, 25, "f32")
        e.task_attach(1, 0, i, 0)
        e.task_reset()
        es.append(e)
    # one-thread figure on a slice (mirrors the reference's one-simulator-per-env design)
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < single_s:
        for _ in range(200):
            es[0].task_step(None)
        n1 += 200
    one = n1 / (time.perf_counter() - t0)
    chunk, done = 100, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        O.vec_task_step(es, chunk)
        done += chunk
    dt = time.perf_counter() - t0
    return {"value": envs * done / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{envs} envs x {done} fused VSS-v0 steps, CPU oracle (float), OpenMP over envs",
            "single_thread_value": one}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--mode", choices=["step", "rollout"], default="step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the extra one-launch rollout leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ensure_built(local_rank)

    import torch
    import torch.distributed as dist
    from rsoccer_amd import _lib as L
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torchrun with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the step engine has no CPU path")
    torch.cuda.set_device(local_rank)
    # RSX_BENCH_FORCE_DIST=1 runs the RCCL path even with one rank (used to test it on a 1-GPU box)
    distributed = world > 1 or os.environ.get("RSX_BENCH_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    B, K, W = args.envs, args.steps, args.warmup
    sim = L.Sim(L.KIND_VSS, 0, 3, 3, 25, B, local_rank)
    sim.task_attach(L.TASK_VSS_V0, seed=0, env_id_base=rank * B, max_episode_steps=0)
    tens = sim.task_tensors()
    stream = torch.cuda.current_stream().cuda_stream
    sim.task_reset(stream)
    mbuf = torch.zeros(L.N_METRICS, dtype=torch.int64, device="cuda")
    ALLREDUCE_EVERY = 100   # steps between metrics all-reduces

    def allreduce_metrics():
        """The only inter-GPU exchange: sum of the 8-entry int64 metrics vector (64 bytes).
        Enqueued on the compute stream as a stream-ordered (host-asynchronous) collective:
        measured on MI355X, recording a HIP event on a busy stream costs ~200 us of stream time,
        so the async_op=True / side-stream forms of torch.distributed (which record events) are
        3 orders of magnitude more expensive than the ~20 us collective itself."""
        mbuf.copy_(tens["metrics"], non_blocking=True)
        dist.all_reduce(mbuf)

    def run(n, timed_mode):
        if timed_mode == "rollout":
            sim.task_rollout(n, stream)
            return
        done = 0
        while done < n:
            m = min(ALLREDUCE_EVERY, n - done)
            sim.task_step_n(m, stream)   # m launches, one per env.step()
            done += m
            if distributed and not os.environ.get("RSX_BENCH_NO_ALLREDUCE"):
                allreduce_metrics()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(mode):
        run(W, mode)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        run(K, mode)
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)
        if distributed:
            t = torch.tensor([wall], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, dev_ms

    wall, dev_ms = timed(args.mode)
    wall_r, dev_ms_r = timed("rollout") if args.mode == "step" and not args.no_rollout else (None, None)

    metrics = sim.read_metrics()
    if distributed:
        mt = torch.from_numpy(metrics).cuda()
        dist.all_reduce(mt)
        metrics = mt.cpu().numpy()

    if rank == 0:
        value = world * B * K / wall
        launch_us = dev_ms * 1e3 / K if args.mode == "step" else dev_ms * 1e3   # per kernel launch
        units_per_launch = B if args.mode == "step" else B * K
        achieved = ALGO_BYTES_PER_ENV_STEP * units_per_launch / (launch_us * 1e-6) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tp) and B == 4096 and args.mode == "step":   # the counters were collected on this configuration
            try:
                traffic = json.load(open(tp)).get("bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "env-steps/sec (whole node), VSS-v0 3v3 @4096 envs, 1/2/4/8 GPU + CPU ref",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"VSS-v0 3v3 fused step, {B} envs per GPU, device-side random actions "
                                   f"(BASELINE.json configs[1]); one kernel launch per env.step()"
                                   if args.mode == "step" else
                                   f"VSS-v0 3v3 fused step, {B} envs per GPU, random actions, all {K} steps in one launch",
                       "envs_per_gpu": B, "launch_mode": args.mode, "sub_steps": 5, "time_step_ms": 25},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("rsx::vss_epl_kernel<%d>" if B >= 131072 and os.environ.get("RSX_LAYOUT") != "lanes" or os.environ.get("RSX_LAYOUT") == "epl"
                                    else "rsx::task_step_kernel<0, 8, 1, 6, %d>") % (0 if args.mode == "step" else 3),
                         "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
                         "avg_launch_us": launch_us},
            "episodes": int(metrics[1]), "env_steps_counted": int(metrics[0]),
        }
        if wall_r is not None:
            line["rollout"] = {"value": world * B * K / wall_r, "unit": "env-steps/s",
                               "us_per_step": dev_ms_r * 1e3 / K,
                               "note": "same K fused steps inside ONE launch (state stays in registers)"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B)
        print(json.dumps(line), flush=True)
    sim.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
