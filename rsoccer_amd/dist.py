"""Multi-GPU use: envs are independent, so a population is sharded by contiguous global env
ids — rank r owns ``[r * envs_per_rank, (r + 1) * envs_per_rank)`` — with no data-path
collective.  The only exchange is a sum of the 8-entry int64 metrics vector
(``torch.distributed`` all-reduce; backend ``nccl`` = RCCL on ROCm, ``gloo`` on CPU)."""
import os

import numpy as np


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard(total_envs, rank=None, world=None):
    """(env_id_base, count) of this rank for a population of ``total_envs`` envs; the remainder
    is spread over the first ranks, every env id belongs to exactly one rank."""
    if rank is None or world is None:
        rank, world = rank_world()
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    q, r = divmod(int(total_envs), world)
    count = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return base, count


def init_process_group(backend=None):
    """torch.distributed bootstrap from the torchrun environment (MASTER_ADDR defaults to
    127.0.0.1).  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    rank, world = rank_world()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def allreduce_metrics(metrics, device=None):
    """Sum an int64 metrics vector over all ranks; returns a numpy array (identity when not
    distributed)."""
    import torch
    import torch.distributed as dist
    m = np.asarray(metrics, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return m.copy()
    t = torch.from_numpy(m.copy())
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def device_tag(device_index):
    """'cuda:3 pci 0000:c5:00.0 <name>' of a visible device — for the per-rank diagnostics of a multi-GPU run."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        pci = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
        return f"cuda:{device_index} pci {pci} {p.name}"
    except Exception as ex:   # diagnostics must never take a run down
        return f"cuda:{device_index} ({ex!r})"


class MetricsCollective:
    """The one inter-GPU exchange of a sharded run: the sum of a small int64 vector (64 bytes every 100 steps),
    off the critical path of every rank.  It must therefore never be able to take the run down:

      * a gloo group over 127.0.0.1 is the control plane (always there: rendezvous, agreement, fallback);
      * the RCCL group (``backend="nccl"``) is created next to it and PROBED — one all-reduce of a vector of ones on
        a side stream, waited for with a timeout from a watchdog thread;
      * the ranks then agree over gloo (MIN of their ok flags): all ok -> RCCL carries the metrics; otherwise every rank
        uses gloo for the 64 bytes and ``describe()`` says why ("gloo (rccl failed: ...)"), so that the scaling
        value of the run still prints.

    ``prefer``: "nccl" (default on a GPU run), "gloo" (CPU tests, shared-device test mode).  ``simulate_failure``
    makes the probe fail on the given ranks (tests of the degraded mode: "all", or a rank number)."""

    def __init__(self, rank, world, device=None, prefer="nccl", timeout_s=60.0, simulate_failure=None, log=None):
        import datetime
        import torch
        import torch.distributed as dist
        self.rank, self.world, self.device = rank, world, device
        self.torch, self.dist = torch, dist
        self.log = log or (lambda msg: None)
        self.reason = None
        self.rccl_ranks = 0
        self.group = None   # RCCL group when it works
        # The control plane is a gloo group that THIS object owns.  When the caller already runs a process group (of any
        # backend: an RCCL world would reject the CPU tensors used below) it gets a second, dedicated gloo group over the
        # same ranks; otherwise the world is initialised here (and torn down in close(), which never touches a group
        # this object did not make).
        self._owns_world = False
        self._made_groups = []
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                if world > 1:
                    raise RuntimeError("MetricsCollective: MASTER_PORT is not set (a fixed default would collide when two "
                                       "runs share a node); launch the ranks with torchrun / `bench.py --gpus N`, which "
                                       "pick a free port, or export one")
                os.environ["MASTER_PORT"] = str(_free_port())
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(60.0, 4 * timeout_s)))
            self._owns_world = True
            self.ctl = dist.group.WORLD
        else:
            self.ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=max(60.0, 4 * timeout_s)))
            self._made_groups.append(self.ctl)
        ok, why = 0, None
        if prefer == "nccl":
            # a failing RCCL collective must end in the probe's timeout, not in torch's watchdog tearing the process
            # down: the two switches are read when the group is made, so they are set for the probe only and restored
            saved = {k: os.environ.get(k) for k in ("TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCH_NCCL_ENABLE_MONITORING")}
            for k in saved:
                os.environ.setdefault(k, "0")
            try:
                ok, why = self._probe_rccl(timeout_s, simulate_failure)
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        else:
            why = f"not requested (prefer={prefer})"
        flags = torch.tensor([ok, 1 - ok], dtype=torch.int64)
        dist.all_reduce(flags, group=self.ctl)               # [ranks ok, ranks failed]
        self.rccl_ranks = int(flags[0])
        if prefer == "nccl" and int(flags[1]) > 0:
            failed = int(flags[1])
            self.reason = why if why else f"{failed} other rank(s) failed the probe"
            if self.group is not None:
                self.log(f"RCCL probe ok here, but {failed} rank(s) failed: the metrics go over gloo on every rank")
            self.group = None
            self.rccl_ranks = 0
        elif prefer != "nccl":
            self.reason = None
        self.backend = "rccl" if self.group is not None else "gloo"

    def _probe_rccl(self, timeout_s, simulate_failure):
        import datetime
        import threading
        torch, dist = self.torch, self.dist
        sim = simulate_failure is not None and (str(simulate_failure) == "all" or str(simulate_failure) == str(self.rank))
        result = {}
        # dist.new_group is a collective: either every rank enters it or none does.  The ranks therefore agree over
        # the control plane FIRST whether all of them can try at all (a device is visible, no all-rank simulated
        # failure) — a rank that cannot would otherwise leave the others inside new_group until the timeout.
        can = 1 if (torch.cuda.is_available() and not (sim and str(simulate_failure) == "all")) else 0
        agree = torch.tensor([can], dtype=torch.int64)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN, group=self.ctl)
        if int(agree[0]) == 0:
            why = ("simulated (--simulate-rccl-failure)" if sim else "no HIP device visible") if not can else "another rank cannot use RCCL"
            self.log(f"RCCL not attempted: {why}")
            return 0, why

        def attempt():
            try:
                if sim:
                    raise RuntimeError("simulated (--simulate-rccl-failure)")
                if not torch.cuda.is_available():
                    raise RuntimeError("no HIP device visible")
                g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s),
                                   device_id=torch.device("cuda", self.device) if self.device is not None else None)
                side = torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(side):
                    t = torch.ones(8, dtype=torch.int64, device=torch.device("cuda", self.device))
                    dist.all_reduce(t, group=g)
                    side.synchronize()
                if int(t[0].item()) != self.world:
                    raise RuntimeError(f"probe all-reduce returned {int(t[0].item())}, expected {self.world}")
                result["group"] = g
            except Exception as ex:   # noqa: BLE001 — anything RCCL throws ends in the degraded mode
                result["error"] = f"{type(ex).__name__}: {ex}".splitlines()[0][:300]

        # dist.new_group is itself a collective over the control plane: every rank must call it, also the ones that
        # simulate a failure afterwards — so a simulated failure skips the call on EVERY rank only with "all";
        # a single simulated rank creates the group and then refuses to use it
        if sim and str(simulate_failure) != "all":
            sim = False
            refuse = True
        else:
            refuse = False
        th = threading.Thread(target=attempt, daemon=True)
        th.start()
        th.join(timeout_s + 5.0)
        if th.is_alive():
            self.log(f"RCCL probe did not finish within {timeout_s:.0f} s")
            return 0, f"probe timed out after {timeout_s:.0f} s"
        if refuse and "group" in result:
            self.log("RCCL probe: simulated failure on this rank")
            return 0, "simulated (--simulate-rccl-failure)"
        if "group" in result:
            self.group = result["group"]
            self._made_groups.append(self.group)
            return 1, None
        self.log(f"RCCL unavailable: {result.get('error')}")
        return 0, result.get("error", "unknown error")

    def describe(self):
        if self.backend == "rccl":
            return "rccl"
        return "gloo" if self.reason is None else f"gloo (rccl failed: {self.reason})"

    def buffer_device(self):
        """where the vectors handed to all_reduce() have to live"""
        return self.torch.device("cuda", self.device) if self.backend == "rccl" else self.torch.device("cpu")

    def all_reduce(self, t, op=None):
        op = op if op is not None else self.dist.ReduceOp.SUM
        self.dist.all_reduce(t, op=op, group=self.group if self.backend == "rccl" else self.ctl)
        return t

    def barrier(self):
        if self.backend == "rccl":
            # a collective on the device: ranks leave it when every rank's stream has reached it
            t = self.torch.zeros(1, dtype=self.torch.int32, device=self.buffer_device())
            self.dist.all_reduce(t, group=self.group)
            self.torch.cuda.synchronize()
        else:
            if self.torch.cuda.is_available():
                self.torch.cuda.synchronize()
            self.dist.barrier(group=self.ctl)

    def close(self):
        """Leaves the process as it was found: the groups made here are destroyed, the world only if this object
        initialised it."""
        try:
            self.dist.barrier(group=self.ctl)
        except Exception:   # noqa: BLE001
            pass
        for g in reversed(self._made_groups):
            try:
                self.dist.destroy_process_group(g)
            except Exception:   # noqa: BLE001
                pass
        self._made_groups = []
        self.group = None
        if self._owns_world:
            try:
                self.dist.destroy_process_group()
            except Exception:   # noqa: BLE001
                pass
            self._owns_world = False
