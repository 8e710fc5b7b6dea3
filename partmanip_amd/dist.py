"""Data-parallel glue: one process per GPU, `torch.distributed` ("nccl" = RCCL over xGMI on
ROCm; "gloo" for the CPU tests).  The reference has no distributed path; the
learner shards by env -- rank r owns envs [r*N/W, (r+1)*N/W) and its own (T, N/W, .) storage,
no rollout data ever moves -- and the only exchange is ONE all-reduce per optimiser step of
the flat gradient buffer with the step's scalars (loss, kl) riding in its tail.

xGMI is point-to-point (7 links x ~153 GB/s per GPU); the message here is 1.2-2.2 MB fp32, i.e.
latency-bound, so everything that must be reduced for a step is packed into that single call.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style env vars; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # debugging aids for 1-GPU boxes: PARTMANIP_DIST_BACKEND=gloo lets several ranks share one device
    # (PARTMANIP_SHARE_GPU=1 maps every rank to cuda:0); RCCL itself needs one GPU per rank.
    backend = os.environ.get("PARTMANIP_DIST_BACKEND") or backend
    if os.environ.get("PARTMANIP_SHARE_GPU") == "1":
        local = 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)      # binds the communicator to this rank's GPU up front
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_envs(num_envs, rank, world):
    """Contiguous env range owned by `rank`."""
    if num_envs % world != 0:
        raise ValueError(f"num_envs={num_envs} is not divisible by world_size={world}")
    per = num_envs // world
    return rank * per, (rank + 1) * per


class GradSync:
    """Mean-reduction of flat buffers across the data-parallel group."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised process group (see init_from_env)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.events = None                 # bench.py: HIP events around every all-reduce (time_collectives)
        self.calls = 0

    def time_collectives(self, on=True):
        """Bracket every all-reduce with HIP events on the issuing stream (bench.py's comm_ms_per_step).  Off by default;
        skipped while a stream is being captured into a hipGraph (events cannot be timed inside a graph)."""
        self.events = [] if on else None
        self.calls = 0

    def comm_ms(self):
        """(total ms, calls) of the bracketed all-reduces since time_collectives(); synchronises."""
        if not self.events:
            return 0.0, self.calls
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.events), self.calls

    def _all_reduce(self, flat):
        self.calls += 1
        if self.events is None or not flat.is_cuda or torch.cuda.is_current_stream_capturing():
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        e1.record()
        self.events.append((e0, e1))

    def mean_(self, flat):
        """In-place average of a flat tensor over ranks (sum + scale: gloo has no AVG op)."""
        self._all_reduce(flat)
        flat.mul_(1.0 / self.world)
        return flat

    def sum_(self, flat):
        self._all_reduce(flat)
        return flat

    def moments_sync(self, mom2, count):
        """All-reduce {sum, sumsq} (fp64) of an advantage batch; returns the global count."""
        self.sum_(mom2)
        return count * self.world

    def any_(self, flag, views_device=None):
        """True on every rank if `flag` is true on ANY rank (MAX all-reduce of one int; a host read -- control flow that all
        ranks must take together, e.g. whether a hipGraph capture succeeded everywhere)."""
        dev = views_device if (views_device is not None and dist.get_backend(self.group) == "nccl") else "cpu"
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def broadcast_(self, t, src=0):
        """Rank `src`'s tensor to every rank (initial parameters / optimiser state: one model, W replicas)."""
        dist.broadcast(t, src=src, group=self.group)
        return t

    def barrier(self):
        dist.barrier(group=self.group)


def maybe_sync():
    """GradSync when a process group with more than one rank exists, else None.  PARTMANIP_FORCE_SYNC=1 also returns
    one for a single-rank group, so that the collective code path (and RCCL itself) can be exercised on a 1-GPU box."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size() > 1 or os.environ.get("PARTMANIP_FORCE_SYNC") == "1":
        return GradSync()
    return None


def resolve_seed(seed_fn):
    """Run `seed_fn()` on rank 0 and hand its (picklable) result to every rank: a run started with `seed: -1` must not
    draw a different seed -- i.e. different initial weights and a different checkpoint directory -- per rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seed_fn()
    box = [seed_fn() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]
