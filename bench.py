#!/usr/bin/env python
"""Headline benchmark: env-steps/s of the fused VSS-v0 3v3 step at 4096 envs per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode step|rollout] [--envs B]

`--gpus N` with N > 1 starts its own N ranks (one process per GPU, `torch.distributed.run` on
127.0.0.1) when it is not already running under torchrun; the torchrun form works as well:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batched env.step() of every env on the rank (BASELINE.json configs[1]:
"VSS-v0 3v3, 4096 batched envs on 1xMI355X, random actions"): device-side random agent action,
OU noise for the other five robots, 5 physics sub-steps, observation, reward, done, TimeLimit
and same-step auto-reset, in ONE kernel launch per step (mode `step`, the default and the value
reported).  All inputs are resident in HBM before the timed region.  With N > 1 each rank owns
envs [rank*B, (rank+1)*B) (weak scaling, no data-path collective); a 64-byte metrics vector is
all-reduced over RCCL every 100 steps (a ~20 us stream-ordered collective).

Prints ONE JSON line on rank 0 (see the driver contract).  Besides the contract's keys:
  roofline      dominant kernel of the timed leg, HIP-event launch average, algorithmic bytes
  steady        the same per-step launches over >= 2000 steps after >= 200,
                whatever --steps / --warmup were
  rollout       the same K steps inside ONE launch
  sweep         (N = 1) per-step and one-launch legs at 65 536 / 1 048 576 / 4 194 304 envs with their
                own roofline fractions: where the path is bandwidth-bound
  configs       (N = 1) BASELINE.json configs[2] (SSLStaticDefenders 1v6 @ 2048) and configs[3] (SSL 11v11
                @ 1024, spread and crowded line-ups: the scrimmage task, every robot commanded on device),
                each also at a bandwidth-bound batch, with SURVEY.md 8(d)'s algorithmic bytes
  python_layer  (N = 1) rate of the Python API on top of the C-ABI, and the reference's Python-layer
                ceiling restated from SURVEY.md
  cpu_baseline  (N = 1) the CPU oracle on this box's host cores, bounded sample
  evidence      (N = 1, LAST key) compact digest of the large-batch legs: [us per step, nominal roofline fraction,
                counter traffic / time over 6.29 TB/s] per leg, configs[2] / configs[3] step times, the policy-loop rates
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 541   # VSS-v0 fused = 2*164 (state r/w) + 48 (cmds) + 160 (obs) + 4 + 1
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
METRIC = "env-steps/sec (whole node), VSS-v0 3v3 @4096 envs, 1/2/4/8 GPU + CPU ref"
ALLREDUCE_EVERY = 100           # steps between metrics all-reduces
STEADY_STEPS, STEADY_WARMUP = 2000, 200
LARGE_WARMUP = 200             # untimed steps before the large-batch legs of the registered tasks: steady state; the first ~150 steps
                               # after a reset, with every env in the same early phase of its first episode, run 3-14 % slower (profiles/r05_leg_windows.txt)
EVENT_MIN_STEPS = 200           # shorter timed regions are measured by the wall clock alone (see timed())
SWEEP_ENVS = (65536, 1048576, 4194304)
HBM_ACHIEVABLE_GBS = 6290.0     # MI355X_MICROARCH.md: what a streaming kernel reaches of the 8 TB/s


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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves (one process per
    GPU, rendezvous on 127.0.0.1) and hand their exit status back.  Rank 0's JSON line is the only
    thing the children write to stdout."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


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
            "single_thread_value": one,
            "note": "a port of this project's own 2-D model (oracle/), not rc-robosim: the reference's physics "
                    "engine is not in /root/reference and cannot be built or run here"}


def kernel_name(layout, mode):
    """kernel symbol of a VSS-v0 3v3 handle from the layout the LIBRARY reports (rsx_task_layout): no threshold is
    restated here"""
    if layout == "one-lane-per-env":
        return "rsx::vss_epl_kernel<0>" if mode == "step" else "rsx::vss_epl_rollout_kernel"
    lanes = {"8-lanes-per-env": 8, "16-lanes-per-env": 16, "32-lanes-per-env": 32, "64-lanes-per-env": 64}.get(layout, 8)
    return "rsx::task_step_kernel<0, %d, 1, 6, %d>" % (lanes, 0 if mode == "step" else 3)


_LEG_TRAFFIC = None


def leg_traffic(leg):
    """(bytes per launch, source) of a large-batch leg from the newest profiles/r*_leg_traffic.json (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of tools/prof_leg_traffic.sh: the counters need their own runs), or (None, None)"""
    global _LEG_TRAFFIC
    if _LEG_TRAFFIC is None:
        import glob
        import re
        _LEG_TRAFFIC = ({}, None)
        files = glob.glob(os.path.join(ROOT, "profiles", "r*_leg_traffic.json"))
        files.sort(key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1)), reverse=True)
        for f in files:
            try:
                doc = json.load(open(f))
                fa = doc.get("factors")
                how = (f"FETCH_SIZE x {fa['FETCH_SIZE']:g} + WRITE_SIZE x {fa['WRITE_SIZE']:g}, factors calibrated in {fa['source']}" if fa
                       else "FETCH_SIZE x 2 + WRITE_SIZE x 1, uncalibrated")
                _LEG_TRAFFIC = (doc.get("legs", {}), os.path.relpath(f, ROOT) + f" [{how}]")
                break
            except Exception:
                continue
    legs, src = _LEG_TRAFFIC
    rec = legs.get(leg)
    if not rec or not rec.get("bytes_per_launch"):
        return None, None
    return rec["bytes_per_launch"], src


def with_traffic(rec, leg, us_per_launch):
    """adds the counter traffic of a leg and what it means in bandwidth: traffic / time against the ~6.29 TB/s a
    streaming kernel reaches (the roofline fraction beside it is SURVEY's algorithmic bytes against the nominal 8)"""
    t, src = leg_traffic(leg)
    rec["traffic"] = t
    if t:
        rec["traffic_source"] = src + " (separate rocprofv3 --pmc passes of this leg, not measured in this run)"
        rec["traffic_over_algorithmic"] = t / (rec["algorithmic_bytes_per_env_step"] * rec["envs"]) if rec.get("algorithmic_bytes_per_env_step") else None
        rec["real_tb_per_s"] = t / (us_per_launch * 1e-6) / 1e12
        rec["real_frac_of_achievable"] = t / (us_per_launch * 1e-6) / 1e9 / HBM_ACHIEVABLE_GBS
    return rec


def roofline_of(envs, launch_us, units_per_launch, mode, traffic=None, traffic_source=None, layout="8-lanes-per-env"):
    achieved = ALGO_BYTES_PER_ENV_STEP * units_per_launch / (launch_us * 1e-6) / 1e9
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
         "kernel": kernel_name(layout, mode), "layout": layout, "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
         "avg_launch_us": launch_us}
    if traffic_source:
        r["traffic_source"] = traffic_source
    if envs < 65536:
        waves = (envs + 7) // 8
        r["regime"] = (f"latency / instruction-issue bound at this batch: {waves} single-wave workgroups on 1024 SIMDs; "
                       "the HBM roofline is the nominal bound of the path (see `sweep` for the bandwidth-bound batches)")
    return r


def rank_logger(rank, world, tag=""):
    """per-rank diagnostics go to stderr with a prefix (stdout carries rank 0's JSON line only)"""
    def log(msg):
        print(f"[bench rank {rank}/{world}{' ' + tag if tag else ''}] {msg}", file=sys.stderr, flush=True)
    return log


def efficiency_vs_n1(value, world):
    """value / (N x the N=1 value) — only when the caller supplies the single-GPU value of the SAME build and box
    (RSX_BENCH_N1_VALUE); a figure read from a checked-in profile would mix runs.  The driver computes scaling
    efficiency itself from its own N = 1, 2, 4, 8 runs."""
    try:
        n1 = float(os.environ["RSX_BENCH_N1_VALUE"])
    except (KeyError, ValueError):
        return None
    if not n1 or world < 1:
        return None
    return {"efficiency_vs_n1": value / (world * n1), "n1_value": n1, "n1_source": "RSX_BENCH_N1_VALUE"}


def dry_run(args, rank, world):
    """Launcher / collective plumbing without a GPU (CPU test of `--gpus N`): the ranks shard the env ids, set up
    the metrics collective exactly like a real run (RCCL cannot come up without a device, so this exercises the
    degraded mode: gloo carries the 64 bytes and the line says why), all-reduce a metrics vector and rank 0 prints
    the line."""
    import torch
    from rsoccer_amd import dist as rdist
    log = rank_logger(rank, world, "dry-run")
    coll = None
    if world > 1:
        coll = rdist.MetricsCollective(rank, world, device=None, prefer="nccl", timeout_s=args.rccl_timeout,
                                       simulate_failure=args.simulate_rccl_failure, log=log)
    base, count = rdist.shard(world * args.envs, rank, world)
    m = torch.zeros(8, dtype=torch.int64)
    m[0] = count * args.steps
    m[7] = base
    if coll:
        coll.all_reduce(m)
        coll.barrier()
    if rank == 0:
        line = {"metric": METRIC, "dry_run": True, "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "env_steps_counted": int(m[0]), "env_id_bases_sum": int(m[7])}
        if coll:
            line["collective"] = {"backend": coll.describe(), "degraded": coll.reason is not None, "ranks": world, "rccl_ranks": coll.rccl_ranks}
        print(json.dumps(line), flush=True)
    if coll:
        coll.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--mode", choices=["step", "rollout"], default="step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the extra one-launch rollout leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the steady / sweep / python_layer legs")
    ap.add_argument("--dry-run", action="store_true", help="launcher + collective plumbing only (no GPU)")
    ap.add_argument("--simulate-rccl-failure", default=None, metavar="all|RANK",
                    help="make the RCCL probe fail (on every rank, or on one): tests the degraded mode (metrics over gloo)")
    ap.add_argument("--rccl-timeout", type=float, default=60.0, help="seconds the RCCL probe may take before gloo takes over")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)
    ensure_built(local_rank)

    import torch
    from rsoccer_amd import _lib as L
    from rsoccer_amd import dist as rdist
    log = rank_logger(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the step engine has no CPU path")
    ndev = torch.cuda.device_count()
    # RSX_BENCH_SHARE_DEVICE=1 (test mode): more ranks than devices, metrics all-reduced over gloo.
    # RCCL wants one device per rank, which is also the only configuration that is a benchmark.
    share = os.environ.get("RSX_BENCH_SHARE_DEVICE") == "1"
    if world > ndev and not share:
        raise SystemExit(f"--gpus {world} but only {ndev} device(s) visible")
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    dev_tag = rdist.device_tag(dev)
    # RSX_BENCH_FORCE_DIST=1 runs the RCCL path even with one rank (used to test it on a 1-GPU box)
    distributed = world > 1 or os.environ.get("RSX_BENCH_FORCE_DIST") == "1"
    coll = None
    if distributed:
        log(f"{dev_tag}; {ndev} device(s) visible; HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}")
        coll = rdist.MetricsCollective(rank, world, device=dev, prefer="gloo" if share else "nccl",
                                       timeout_s=args.rccl_timeout, simulate_failure=args.simulate_rccl_failure, log=log)
        log(f"metrics collective: {coll.describe()} ({coll.rccl_ranks}/{world} ranks on RCCL)")
    on_rccl = coll is not None and coll.backend == "rccl"

    B, K, W = args.envs, args.steps, args.warmup
    sim = L.Sim(L.KIND_VSS, 0, 3, 3, 25, B, dev)
    sim.task_attach(L.TASK_VSS_V0, seed=0, env_id_base=rank * B, max_episode_steps=0)
    tens = sim.task_tensors()
    layout = sim.task_layout()
    stream = torch.cuda.current_stream().cuda_stream
    sim.task_reset(stream)
    mbuf = torch.zeros(L.N_METRICS, dtype=torch.int64, device=coll.buffer_device() if coll else "cpu")

    def allreduce_metrics():
        """The only inter-GPU exchange: sum of the 8-entry int64 metrics vector (64 bytes), all eight
        entries counted on the device.  Enqueued on the compute stream as a stream-ordered
        (host-asynchronous) collective: measured on MI355X, recording a HIP event on a busy stream
        costs ~200 us of stream time, so the async_op=True / side-stream forms of torch.distributed
        (which record events) are 3 orders of magnitude more expensive than the ~20 us collective."""
        sim.metrics_fold(stream)
        mbuf.copy_(tens["metrics"], non_blocking=on_rccl)   # degraded mode (gloo): a synchronous 64-byte read-back
        coll.all_reduce(mbuf)

    stepped = [0]   # per-step launches of `sim` so far: the all-reduce falls on every ALLREDUCE_EVERY-th of them

    def run(s, n, timed_mode):
        if timed_mode == "rollout":
            s.task_rollout(n, stream)
            return
        done = 0
        while done < n:
            m = min(ALLREDUCE_EVERY - stepped[0] % ALLREDUCE_EVERY if s is sim else ALLREDUCE_EVERY, n - done)
            s.task_step_n(m, stream)   # m launches, one per env.step()
            done += m
            if s is sim:
                stepped[0] += m
                if distributed and stepped[0] % ALLREDUCE_EVERY == 0 and not os.environ.get("RSX_BENCH_NO_ALLREDUCE"):
                    allreduce_metrics()

    def barrier():
        if distributed:
            coll.barrier()
        torch.cuda.synchronize()

    def timed(s, n, warm, mode):
        """warm untimed steps, then EXACTLY n steps bracketed by barrier + synchronize; returns the
        max wall time over ranks and this rank's HIP-event time of the same region (ms).  Every rank starts its
        clock when the opening barrier releases it and stops it when ITS n steps are complete on the device; the
        maximum over ranks is then the time until the last rank was done — the closing barrier itself (an RCCL
        collective of ~50-100 us, comparable to a short timed region) is not part of any rank's n steps."""
        run(s, warm, mode) if warm else None
        barrier()
        # HIP events bracket the launches of regions of >= EVENT_MIN_STEPS launches only: recording the pair costs ~14 us of
        # stream time (tools/exp_sync_latency.py: 20 launches 190 us without, 204 us with), 7 % of the driver's --steps 20
        # region and nothing of the 2000-launch leg `roofline` is computed from; a short region reports its wall clock
        use_events = mode == "rollout" or n >= EVENT_MIN_STEPS
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if use_events:
            ev0.record()
        run(s, n, mode)
        if use_events:
            ev1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        barrier()
        dev_ms = ev0.elapsed_time(ev1) if use_events else wall * 1e3
        if distributed:
            t = torch.tensor([wall], dtype=torch.float64, device=mbuf.device)
            coll.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            wall = float(t.item())
        return wall, dev_ms

    wall, dev_ms = timed(sim, K, W, args.mode)
    per_rank_ms = None
    if distributed:
        g = torch.zeros(world, dtype=torch.float64, device=mbuf.device)
        g[rank] = dev_ms / K
        coll.all_reduce(g)
        per_rank_ms = [float(x) for x in g.cpu()]
        tags = [None] * world
        torch.distributed.all_gather_object(tags, dev_tag, group=coll.ctl)
    extra = not args.no_extra and args.mode == "step"
    steady = None
    if extra and (K < STEADY_STEPS or W < STEADY_WARMUP):
        steady = timed(sim, STEADY_STEPS, max(0, STEADY_WARMUP - (W + K)), "step")
    roll = timed(sim, K, W, "rollout") if args.mode == "step" and not args.no_rollout else None

    metrics = sim.read_metrics()
    if distributed:
        mt = torch.from_numpy(metrics).to(mbuf.device)
        coll.all_reduce(mt)
        metrics = mt.cpu().numpy()

    line = None
    if rank == 0:
        value = world * B * K / wall
        launch_us = dev_ms * 1e3 / K if args.mode == "step" else dev_ms * 1e3   # per kernel launch
        units_per_launch = B if args.mode == "step" else B * K
        traffic, tsrc = None, None
        import glob
        import re
        for tp in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True,
                         key=lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1))):   # newest ROUND first (r10 after r9)
            if B == 4096 and args.mode == "step":   # the counters were collected on this configuration
                try:
                    doc = json.load(open(tp))
                    traffic = doc.get("bytes_per_launch")
                    fa = doc.get("factors")
                    tsrc = (os.path.relpath(tp, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this command; "
                            + (f"FETCH_SIZE x {fa['FETCH_SIZE']:g} + WRITE_SIZE x {fa['WRITE_SIZE']:g}, factors calibrated in {fa['source']}" if fa
                               else "FETCH_SIZE x 2 + WRITE_SIZE x 1, uncalibrated") + "; not measured in this run)")
                    break
                except Exception:
                    traffic = None
        line = {
            "metric": METRIC,
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall * 1e3 / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"VSS-v0 3v3 fused step, {B} envs per GPU, device-side random actions "
                                   f"(BASELINE.json configs[1]); one kernel launch per env.step()"
                                   if args.mode == "step" else
                                   f"VSS-v0 3v3 fused step, {B} envs per GPU, random actions, all {K} steps in one launch",
                       "envs_per_gpu": B, "launch_mode": args.mode, "sub_steps": 5, "time_step_ms": 25,
                       "physics": "project-defined 2-D rigid-body model, version 2 (docs/PHYSICS.md, DESIGN.md 4: wall-aware contacts — VSS: held "
                                  "axes + goal posts as chords, since round 6; the headline step was 8.8 us under model v1, 9.4-9.5 under v2); "
                                  "not validated against rc-robosim, whose sources are not part of the reference tree"},
            "episodes": int(metrics[1]), "env_steps_counted": int(metrics[0]),
        }
        # which clock and which window every top-level figure uses (rounds 1-4 printed the steady leg as `value`; since round 5
        # `value` is the flag-defined region, and a region of fewer than EVENT_MIN_STEPS launches is timed by the host clock)
        line["value_clock"] = ("host wall clock around the timed region (barrier + synchronize on both sides), max over ranks"
                               + ("" if K >= EVENT_MIN_STEPS or args.mode == "rollout" else
                                  f"; no HIP events inside a region of {K} < {EVENT_MIN_STEPS} launches (the pair costs ~14 us of stream time)"))
        line["value_window"] = f"{K} steps after {W} warm-up steps (the flags)"
        line["value_timed_region"] = value                     # the same number under the name round 5 introduced it with
        # `value` / `ms_per_step` are ALWAYS the timed region the flags ask for (the contract; comparable across rounds).  A region
        # of fewer than 200 launches (the driver's --steps 20 is 0.2 ms of wall clock: +-10 % from run to run, early-episode
        # steps) is a noisy sample of the steady rate, which is published beside it.
        if args.mode == "step" and steady is not None:
            line["value_steady"] = world * B * STEADY_STEPS / steady[0]
            line["ms_per_step_steady"] = steady[0] * 1e3 / STEADY_STEPS
            line["value_steady_source"] = (f"steady leg: {STEADY_STEPS} per-step launches after {max(W + K, STEADY_WARMUP)}, same barrier + synchronize bracket "
                                           "(what `roofline` and the rocprof average of this command describe)")
        # `roofline` is the dominant kernel's launch average over the STEADY leg (>= 2000 launches after >= 200: what a
        # rocprofv3 --kernel-trace --stats of this command averages over, profiles/); the short timed region of the
        # driver's flags (early-episode steps, a few launches) is kept beside it as frac_timed_region
        if args.mode == "step" and steady is not None:
            sd_us = steady[1] * 1e3 / STEADY_STEPS
            line["roofline"] = roofline_of(B, sd_us, B, "step", traffic, tsrc, layout)
            line["roofline"]["source"] = f"steady leg: HIP-event average of {STEADY_STEPS} per-step launches after {max(W + K, STEADY_WARMUP)}"
            line["roofline"]["frac_timed_region"] = roofline_of(B, launch_us, units_per_launch, "step")["frac"]
            line["roofline"]["avg_launch_us_timed_region"] = launch_us
            line["roofline"]["timed_region_clock"] = "HIP events" if K >= EVENT_MIN_STEPS else "wall clock of the bracket / steps (no events in a region this short)"
        else:
            line["roofline"] = roofline_of(B, launch_us, units_per_launch, args.mode, traffic, tsrc, layout)
            line["roofline"]["source"] = (f"timed region: {'HIP-event' if args.mode == 'rollout' or K >= EVENT_MIN_STEPS else 'wall-clock'} average of "
                                          f"{K if args.mode == 'step' else 1} launch(es) after {W} steps")
            if args.mode == "rollout":
                line["roofline"]["notional"] = True
        if distributed:
            line["collective"] = {"backend": "gloo (shared device, test mode)" if share else coll.describe(),
                                  "degraded": bool(coll.reason is not None and not share),   # RCCL was wanted and is not carrying the metrics
                                  "ranks": world, "rccl_ranks": coll.rccl_ranks, "devices": tags,
                                  "payload_bytes": 8 * L.N_METRICS, "every_steps": ALLREDUCE_EVERY,
                                  "allreduces_in_timed_region": (W + K) // ALLREDUCE_EVERY - W // ALLREDUCE_EVERY,
                                  "per_rank_ms_per_step": per_rank_ms}
            eff = efficiency_vs_n1(value, world)
            if eff:
                line.update(eff)
        if steady is not None:
            sw, sd = steady
            line["steady"] = {"value": world * B * STEADY_STEPS / sw, "unit": "env-steps/s", "steps": STEADY_STEPS,
                              "after_steps": max(W + K, STEADY_WARMUP), "us_per_step": sd * 1e3 / STEADY_STEPS,
                              "roofline_frac": roofline_of(B, sd * 1e3 / STEADY_STEPS, B, "step")["frac"],
                              "note": "same per-step launches as `value`, SURVEY.md 8(d) config 2 sample size"}
        elif args.mode == "step":
            line["steady"] = {"value": value, "unit": "env-steps/s", "steps": K, "after_steps": W,
                              "us_per_step": launch_us, "roofline_frac": line["roofline"]["frac"],
                              "note": "the timed leg itself meets SURVEY.md 8(d) config 2 (>= 2000 steps after >= 200)"}
        if roll is not None:
            rw, rd = roll
            line["rollout"] = {"value": world * B * K / rw, "unit": "env-steps/s",
                               "us_per_step": rd * 1e3 / K,
                               "roofline_frac": roofline_of(B, rd * 1e3, B * K, "rollout")["frac"], "notional": True,
                               "note": "same K fused steps inside ONE launch: the state stays in registers and only the last step's "
                                       "observation / reward rows are written, so the algorithmic bytes of K env.step() calls are NOT "
                                       "moved — the fraction is notional (a throughput in roofline units), not bandwidth evidence"}
    sim.close()

    if rank == 0 and world == 1 and extra:
        line["sweep"] = sweep(L, torch, dev, timed)
        line["configs"] = other_configs(L, torch, dev, timed)
        line["python_layer"] = python_layer(torch, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(B)
    if rank == 0 and world == 1 and extra:
        line["evidence"] = evidence(line)   # LAST key: the driver keeps the tail of stdout
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        coll.close()
        if coll.reason is not None and not share:
            # degraded mode: a wedged RCCL communicator may hang the interpreter's teardown — everything is printed
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)


def evidence(line):
    """Compact digest of the legs behind the roofline claims, as the LAST key of the line (a reader of the tail of stdout sees
    it): [us per step, nominal fraction of 8 TB/s (SURVEY bytes), counter traffic / time as a fraction of 6.29 TB/s or null]."""
    r3 = lambda x: None if x is None else round(x, 3)
    ev = {}
    for rec in line.get("sweep", []):
        st = rec.get("step")
        if st and rec["envs"] >= 1 << 20:
            ev[f"vss_{rec['envs'] >> 20}M"] = [round(st["us_per_step"], 1), r3(st["roofline_frac"]), r3(st.get("real_frac_of_achievable"))]
    tags = {"SSLStaticDefenders": "sd", "SSLDribbling": "drib", "SSLContested": "cont", "SSLPassEndurance": "pass"}
    c3 = []
    for rec in line.get("configs", []):
        if "us_per_step" not in rec:
            continue
        w, B = rec["workload"], rec["envs"]
        trip = [round(rec["us_per_step"], 1), r3(rec["roofline_frac"]), r3(rec.get("real_frac_of_achievable"))]
        if w.startswith("configs[2]"):
            ev["configs2_us"] = round(rec["us_per_step"], 2)
        elif w.startswith("configs[3]"):
            c3.append(round(rec["us_per_step"], 2))
        elif "scrimmage" in w and "steps" not in w.split("envs")[-1]:
            ev[("scrimC_" if "crowded" in w else "scrim_") + str(B)] = trip
        elif "scrimmage" in w:
            ev["scrimC_late_" + str(B)] = trip
        else:
            for k, t in tags.items():
                if w.startswith(k) and B >= 1 << 20:
                    ev[f"{t}_{B >> 20}M"] = trip
    if c3:
        ev["configs3_us"] = c3
    pl = line.get("python_layer", {})
    if "graph_policy_loop_env_steps_per_s" in pl:
        ev["graph_loop"] = float("%.3g" % pl["graph_policy_loop_env_steps_per_s"])
        ev["eager_loop"] = float("%.3g" % pl.get("eager_policy_loop_env_steps_per_s", 0.0))
        if "fused_policy_loop_env_steps_per_s" in pl:
            ev["fused_policy_loop"] = float("%.3g" % pl["fused_policy_loop_env_steps_per_s"])
    return ev


def sweep(L, torch, dev, timed, n=100, warm=LARGE_WARMUP):
    """Where the path is bandwidth-bound: the same fused step at large batches, one launch per step
    and all steps in one launch, each with its own roofline fraction (HIP events on the launch stream)."""
    out = []
    for B in SWEEP_ENVS:
        try:
            free, _ = torch.cuda.mem_get_info()
            if free < B * 1200:
                out.append({"envs": B, "skipped": f"only {free >> 20} MiB free"})
                continue
            # two handles, one after the other: at >= 1 M envs the rate depends on where the driver put the arrays (+-5 % from allocation to
            # allocation, profiles/r05_row_stride.txt); the figure is the mean of the two, both are listed
            runs = []
            for _ in range(2 if B >= 1 << 20 else 1):
                s = L.Sim(L.KIND_VSS, 0, 3, 3, 25, B, dev)
                s.task_attach(L.TASK_VSS_V0, seed=0, env_id_base=0, max_episode_steps=0)
                lay = s.task_layout()
                s.task_reset(torch.cuda.current_stream().cuda_stream)
                w1, d1 = timed(s, n, warm, "step")
                w2, d2 = timed(s, n, 0, "rollout")
                s.close()
                del s
                torch.cuda.empty_cache()
                runs.append((w1, d1, w2, d2))
            w1, d1, w2, d2 = (sum(r[i] for r in runs) / len(runs) for i in range(4))
            step = {"us_per_step": d1 * 1e3 / n, "value": B * n / w1, "envs": B,
                    "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
                    "roofline_frac": roofline_of(B, d1 * 1e3 / n, B, "step")["frac"]}
            if len(runs) > 1:
                step["us_per_step_runs"] = [r[1] * 1e3 / n for r in runs]
            with_traffic(step, f"vss:{B}", d1 * 1e3 / n)
            out.append({"envs": B, "kernel": kernel_name(lay, "step"), "layout": lay, "step": step,
                        "rollout": {"us_per_step": d2 * 1e3 / n, "value": B * n / w2, "notional": True,
                                    "roofline_frac": roofline_of(B, d2 * 1e3, B * n, "rollout")["frac"]},
                        "steps": n, "warmup": warm})
        except Exception as ex:   # a failed sweep point must not lose the headline line
            out.append({"envs": B, "error": repr(ex)})
    return out


def other_configs(L, torch, dev, timed):
    """The other single-GPU configurations of BASELINE.json, fused, one launch per step, device-side random
    actions; algorithmic bytes per env-step from SURVEY.md 8(d): state r/w + commands + obs + reward + done."""
    def ssl_bytes(n_robots, obs_dim):
        return 2 * 4 * (5 + 11 * n_robots) + 4 * 8 * n_robots + 4 * obs_dim + 5
    SD = ssl_bytes(7, 24)      # 981
    SC = ssl_bytes(22, 46)     # 2869
    cases = (("configs[2] SSLStaticDefenders-v0 1v6", 1, 2, 1, 6, L.TASK_SSL_STATIC_DEFENDERS, 2048, SD, 2000, 200),
             ("SSLStaticDefenders-v0 1v6, 262 144 envs (one-lane-per-env kernel)", 1, 2, 1, 6, L.TASK_SSL_STATIC_DEFENDERS, 262144, SD, 100, LARGE_WARMUP),
             ("SSLStaticDefenders-v0 1v6, 1 048 576 envs (one-lane-per-env kernel)", 1, 2, 1, 6, L.TASK_SSL_STATIC_DEFENDERS, 1048576, SD, 60, LARGE_WARMUP),
             ("SSLStaticDefenders-v0 1v6, 4 194 304 envs (1.4 GB of state: beyond the 256 MB memory-side cache)", 1, 2, 1, 6,
              L.TASK_SSL_STATIC_DEFENDERS, 4194304, SD, 30, LARGE_WARMUP),
             ("SSLDribbling-v0 1v4, 1 048 576 envs (one-lane-per-env kernel)", 1, 2, 1, 4, L.TASK_SSL_DRIBBLING, 1048576, ssl_bytes(5, 21), 60, LARGE_WARMUP),
             ("SSLContestedPossession-v0 1v1, 1 048 576 envs (one-lane-per-env kernel)", 1, 2, 1, 1, L.TASK_SSL_CONTESTED, 1048576, ssl_bytes(2, 14), 60, LARGE_WARMUP),
             ("SSLPassEndurance-v0 2v0, 1 048 576 envs (one-lane-per-env kernel)", 1, 2, 2, 0, L.TASK_SSL_PASS_ENDURANCE, 1048576, ssl_bytes(2, 16), 60, LARGE_WARMUP),
             ("configs[3] SSL 11v11 division-A, scrimmage task, spread line-up", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE, 1024, SC, 1000, 100),
             ("configs[3] SSL 11v11 division-A, scrimmage task, crowded line-up (worst-case contacts)", 1, 1, 11, 11,
              L.TASK_SSL_SCRIMMAGE_CROWDED, 1024, SC, 1000, 100),
             ("SSL 11v11 scrimmage, spread, 65 536 envs (four lanes per env from 32 768 envs)", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE, 65536, SC, 60, 20),
             ("SSL 11v11 scrimmage, crowded, 65 536 envs (four lanes per env from 65 536 envs)", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE_CROWDED, 65536, SC, 60, 20),
             ("SSL 11v11 scrimmage, spread, 262 144 envs (four lanes per env)", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE, 262144, SC, 30, 10),
             ("SSL 11v11 scrimmage, crowded, 262 144 envs (four lanes per env)", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE_CROWDED, 262144, SC, 30, 10),
             # the two legs above time steps 21-80 / 11-40 after a reset, where the crowded line-up is one scrum per env; later on:
             ("SSL 11v11 scrimmage, crowded, 65 536 envs, steps 201-260 after a reset", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE_CROWDED, 65536, SC, 60, 200),
             ("SSL 11v11 scrimmage, crowded, 262 144 envs, steps 121-150 after a reset", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE_CROWDED, 262144, SC, 30, 120))
    LEG = {L.TASK_SSL_STATIC_DEFENDERS: "sd", L.TASK_SSL_DRIBBLING: "drib", L.TASK_SSL_CONTESTED: "cont",
           L.TASK_SSL_PASS_ENDURANCE: "pass", L.TASK_SSL_SCRIMMAGE: "scrim", L.TASK_SSL_SCRIMMAGE_CROWDED: "scrimC"}
    out = []
    for name, kind, ft, nb, ny, task, B, bytes_, n, warm in cases:
        try:
            free, _ = torch.cuda.mem_get_info()
            if free < B * 1400:
                out.append({"workload": name, "envs": B, "skipped": f"only {free >> 20} MiB free"})
                continue
            s = L.Sim(kind, ft, nb, ny, 25, B, dev)
            s.task_attach(task, seed=0, env_id_base=0, max_episode_steps=0)
            lay = s.task_layout()
            s.task_reset(torch.cuda.current_stream().cuda_stream)
            w, d = timed(s, n, warm, "step")
            s.close()
            del s
            torch.cuda.empty_cache()
            us = d * 1e3 / n
            rec = {"workload": name, "envs": B, "layout": lay, "us_per_step": us, "value": B * n / w, "unit": "env-steps/s",
                   "algorithmic_bytes_per_env_step": bytes_, "roofline_frac": bytes_ * B / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                   "steps": n, "warmup": warm}
            if B >= 65536:
                with_traffic(rec, f"{LEG[task]}:{B}", us)
            out.append(rec)
        except Exception as ex:
            out.append({"workload": name, "envs": B, "error": repr(ex)})
    return out


def python_layer(torch, dev, B=4096, n=2000):
    """Rate of the Python API above the C-ABI (what a trainer calls), with device-resident actions;
    and the reference's own Python-layer ceiling, restated."""
    out = {"reference_python_layer_ceiling_steps_per_s_per_core": 4900,
           "reference_note": "SURVEY.md 0-4: the reference's Python hooks alone (no physics) cap it at ~4.9 k env-steps/s/core"}
    try:
        from rsoccer_amd.vec import VecVSSEnv
        env = VecVSSEnv(B, device=dev, seed=0)
        env.reset()
        act = torch.zeros(B, 2, device=f"cuda:{dev}").uniform_(-1, 1)
        for _ in range(200):
            env.step(act)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            env.step(act)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        env.close()
        out.update({"vec_api_calls_per_s": n / dt, "vec_api_env_steps_per_s": B * n / dt, "vec_api_envs": B,
                    "vec_api_note": "VecVSSEnv.step(actions) with a device-resident [B, 2] action tensor, host-asynchronous"})
    except Exception as ex:
        out["vec_api_error"] = repr(ex)
    # the robosim-shaped pair step(cmds) + get_state() for a batch in the reference's float64 wire format, PCIe both ways included
    # (ABI 6: pinned wire buffers, conversion on the device; never `value`)
    try:
        import numpy as np
        from rsoccer_amd import _lib as L
        sim = L.Sim(L.KIND_VSS, 0, 3, 3, 25, B, dev)
        wire = sim.wire_buffers()
        wire[0][...] = np.random.default_rng(0).uniform(-30, 30, wire[0].shape)
        for _ in range(20):
            sim.step_wire()
        m = 300
        t0 = time.perf_counter()
        for _ in range(m):
            sim.step_wire()
        dt = (time.perf_counter() - t0) / m
        sim.close()
        out.update({"host_wire_format_us_per_step": dt * 1e6, "host_wire_format_env_steps_per_s": B / dt,
                    "host_wire_format_note": f"raw VSS 3v3, {B} envs: rsx_step_wire — float64 commands in / float64 state out through pinned "
                                             "buffers, conversion on the device, one synchronisation per step (rsim.py:102,105 for a batch)"})
    except Exception as ex:
        out["host_wire_format_error"] = repr(ex)
    # policy in the loop (the reference's training loop, README.md:116-133, with a 40-64-2 tanh MLP on the GPU): eager, and
    # policy(obs) -> env.step(actions) captured into a hipGraph and replayed (rsx_task_enable_capture: the step counter that
    # keys the random draws lives on the device, every replay advances it; tests/test_gpu_graph.py is the parity proof)
    try:
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import vec_policy_loop as VPL
        from rsoccer_amd.vec import VecVSSEnv
        with torch.no_grad():
            env = VecVSSEnv(B, device=dev, seed=0)
            env.reset()
            policy = VPL.make_policy(env.sim.obs_dim, env.sim.act_dim, env.device)
            actions = torch.zeros(B, env.sim.act_dim, device=env.device)
            VPL.run_eager(env, policy, actions, 100)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            VPL.run_eager(env, policy, actions, 1000)
            torch.cuda.synchronize()
            out["eager_policy_loop_env_steps_per_s"] = B * 1000 / (time.perf_counter() - t0)
            best = 0.0
            for iters in (1, 8):
                g = VPL.build_graph(env, policy, actions, iters)
                for _ in range(5):
                    g.replay()
                torch.cuda.synchronize()
                reps = 3000 // iters
                t0 = time.perf_counter()
                for _ in range(reps):
                    g.replay()
                torch.cuda.synchronize()
                v = B * reps * iters / (time.perf_counter() - t0)
                out[f"graph_policy_loop_env_steps_per_s_{iters}_per_graph"] = v
                best = max(best, v)
                del g
            out["graph_policy_loop_env_steps_per_s"] = best
            try:   # the same MLP as ONE hand-written kernel (examples/fused_policy.hip): what the loop costs when the policy is one launch
                fpol = VPL.make_fused_policy(env.sim.obs_dim, env.sim.act_dim, env.device)
                VPL.run_eager(env, fpol, actions, 50)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                VPL.run_eager(env, fpol, actions, 2000)
                torch.cuda.synchronize()
                out["eager_fused_policy_loop_env_steps_per_s"] = B * 2000 / (time.perf_counter() - t0)   # two launches per iteration: eager is not host-bound
                for iters in (1, 8):
                    g = VPL.build_graph(env, fpol, actions, iters)
                    for _ in range(5):
                        g.replay()
                    torch.cuda.synchronize()
                    reps = 3000 // iters
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        g.replay()
                    torch.cuda.synchronize()
                    out[f"graph_fused_policy_loop_env_steps_per_s_{iters}_per_graph"] = B * reps * iters / (time.perf_counter() - t0)
                    del g
                out["fused_policy_loop_env_steps_per_s"] = max(out["eager_fused_policy_loop_env_steps_per_s"],
                                                               out["graph_fused_policy_loop_env_steps_per_s_1_per_graph"], out["graph_fused_policy_loop_env_steps_per_s_8_per_graph"])
                out["fused_policy_loop_note"] = ("the same 40-64-2 MLP as ONE hand-written kernel (examples/fused_policy.hip) -> VecVSSEnv.step(actions): two launches "
                                                 "per iteration; eager and replayed from a graph")
            except Exception as ex:
                out["fused_policy_loop_error"] = repr(ex)
            out["policy_loop_note"] = (f"{B} envs, 40-64-2 tanh MLP (two addmm + two tanh kernels) -> VecVSSEnv.step(actions); graph = "
                                       "torch.cuda.CUDAGraph replay of 1 or 8 iterations per graph, five kernel nodes per iteration")
            env.close()
    except Exception as ex:
        out["policy_loop_error"] = repr(ex)
    for env_id, key in (("SSLStaticDefenders-v0", "single_env_ssl_steps_per_s"),):
        try:
            import rsoccer_amd
            env = rsoccer_amd.make(env_id)
            env.reset()
            a = env.action_space.sample()
            for _ in range(50):
                _, _, term, trunc, _ = env.step(a)
                if term or trunc:
                    env.reset()
            m = 500
            t0 = time.perf_counter()
            for _ in range(m):
                _, _, term, trunc, _ = env.step(a)
                if term or trunc:
                    env.reset()
            out[key] = m / (time.perf_counter() - t0)
            env.close()
        except Exception as ex:
            out[key + "_error"] = repr(ex)
    try:
        import rsoccer_amd
        env = rsoccer_amd.make("VSS-v0")
        env.reset()
        a = env.action_space.sample()
        for _ in range(50):
            env.step(a)
        m = 500
        t0 = time.perf_counter()
        for _ in range(m):
            _, _, term, trunc, _ = env.step(a)
            if term or trunc:
                env.reset()
        dt = time.perf_counter() - t0
        env.close()
        out.update({"single_env_steps_per_s": m / dt,
                    "single_env_note": "rsoccer_amd.make('VSS-v0'): the reference-shaped class, Python hooks + one host-format "
                                       "rsx_step (PCIe both ways, synchronous) per step"})
    except Exception as ex:
        out["single_env_error"] = repr(ex)
    return out


if __name__ == "__main__":
    main()
