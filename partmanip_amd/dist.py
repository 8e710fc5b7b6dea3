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


_PROBES = {}                                                   # probes issued per communicator name (the store keys must be fresh)


def shard_envs(num_envs, rank, world):
    """Contiguous env range owned by `rank`."""
    if num_envs % world != 0:
        raise ValueError(f"num_envs={num_envs} is not divisible by world_size={world}")
    per = num_envs // world
    return rank * per, (rank + 1) * per


class CollectiveError(RuntimeError):
    """A collective failed or timed out; the message says which rank / device / communicator / call and what to try next."""


def _where(name, group):
    dev = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu"
    return (f"rank {dist.get_rank()} of {dist.get_world_size()} (device {dev}), communicator '{name}', backend "
            f"{dist.get_backend(group)}")


def _hint(backend):
    return ("re-run with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL (RCCL) and HSA_ENABLE_IPC_MODE_LEGACY=0 exported for every rank; "
            "PARTMANIP_DP_GRAPHS=split keeps the all-reduce out of the hipGraphs, PARTMANIP_OVERLAP=0 uses ONE communicator"
            if backend == "nccl" else "check that every rank reached the same collective (PARTMANIP_DP_GRAPHS / sampler / tricks agree)")


class GradSync:
    """Mean-reduction of flat buffers across the data-parallel group."""

    def __init__(self, group=None, name="default"):
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised process group (see init_from_env)")
        self.group = group
        self.name = name                   # 'actor' / 'critic' / 'student': said in every diagnostic
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.events = None                 # bench.py: HIP events around every all-reduce (time_collectives)
        self.calls = 0
        self.mode = None                   # the learner's launch structure ('capture' / 'split' / 'eager'), for diagnostics only

    def probe(self, device, timeout_s=None):
        """The FIRST collective of this communicator, time-boxed: a 4-element all-reduce issued asynchronously and polled for at
        most `timeout_s` (PARTMANIP_PROBE_TIMEOUT, default 60 s -- RCCL builds its rings / trees inside this call).  Every rank then
        publishes its outcome through the rendezvous STORE (host TCP, independent of the communicator under test) and reads the
        others': returns None when the collective worked everywhere, else a text naming the ranks that failed and why -- the same
        on every rank, so that all of them take the same fallback.  PARTMANIP_TEST_COLLECTIVE_FAIL='rank<r>:<name>' makes rank r's
        probe of communicator <name> report an error after the collective (only the store tells the others); '...:absent' makes
        it skip the collective, so the other ranks run into the time box (tests)."""
        import time
        timeout_s = float(os.environ.get("PARTMANIP_PROBE_TIMEOUT", "60")) if timeout_s is None else timeout_s
        be = dist.get_backend(self.group)
        err = None
        inject = os.environ.get("PARTMANIP_TEST_COLLECTIVE_FAIL")
        mine = f"rank{dist.get_rank()}:{self.name}"
        try:
            if inject == mine + ":absent":                     # this rank never joins: the others time out
                raise RuntimeError("PARTMANIP_TEST_COLLECTIVE_FAIL (absent)")
            t = torch.ones(4, dtype=torch.float32, device=device if be == "nccl" else "cpu")
            work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            t0 = time.monotonic()
            while not work.is_completed():
                if time.monotonic() - t0 > timeout_s:
                    raise TimeoutError(f"no completion within {timeout_s:.0f} s")
                time.sleep(0.002)
            work.wait()
            if float(t.sum().item()) != 4.0 * self.world:
                raise RuntimeError(f"wrong sum {t.tolist()} (expected {self.world} per element)")
            if inject == mine:                                 # this rank alone sees an error: the others learn it over the store
                raise RuntimeError("PARTMANIP_TEST_COLLECTIVE_FAIL")
        except Exception as e:                                  # noqa: BLE001 -- whatever the backend raises is the diagnosis
            err = f"{type(e).__name__}: {str(e)[:200]}"
        # agreement through the store: one key per rank and probe; a rank that never writes its key counts as failed
        store = dist.distributed_c10d._get_default_store()
        _PROBES[self.name] = _PROBES.get(self.name, 0) + 1
        tag = f"partmanip/probe/{self.name}/{_PROBES[self.name]}"
        store.set(f"{tag}/{dist.get_rank()}", err or "ok")
        bad = []
        for r in range(dist.get_world_size()):
            try:
                store.wait([f"{tag}/{r}"], __import__("datetime").timedelta(seconds=timeout_s + 30))
                v = store.get(f"{tag}/{r}").decode()
            except Exception as e:                              # noqa: BLE001
                v = f"rank {r} never reported ({type(e).__name__})"
            if v != "ok":
                bad.append(f"rank {r}: {v}")
        if not bad:
            return None
        msg = (f"first all-reduce of communicator '{self.name}' failed on {len(bad)} of {dist.get_world_size()} ranks [{'; '.join(bad)}] "
               f"-- seen from {_where(self.name, self.group)}, launch structure {self.mode or 'n/a'}; {_hint(be)}")
        if err is not None and be == "nccl":                    # a hung collective of THIS rank: release what can be released
            try:
                pg = self.group if self.group is not None else dist.distributed_c10d._get_default_group()
                getattr(pg, "abort", lambda: None)()
            except Exception:                                   # noqa: BLE001
                pass
        return msg

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
        try:
            if self.events is None or not flat.is_cuda or torch.cuda.is_current_stream_capturing():
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                return
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            e1.record()
            self.events.append((e0, e1))
        except CollectiveError:
            raise
        except Exception as e:                                  # noqa: BLE001 -- say WHICH collective before the stack unwinds
            cap = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
            raise CollectiveError(f"all-reduce #{self.calls} of {flat.numel()} {flat.dtype} elements failed on {_where(self.name, self.group)}"
                                  f"{' while a hipGraph was being captured' if cap else ''}, launch structure {self.mode or 'n/a'}: "
                                  f"{type(e).__name__}: {str(e)[:300]} -- {_hint(dist.get_backend(self.group))}") from e

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


def maybe_sync(name="actor"):
    """GradSync when a process group with more than one rank exists, else None.  PARTMANIP_FORCE_SYNC=1 also returns
    one for a single-rank group, so that the collective code path (and RCCL itself) can be exercised on a 1-GPU box."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size() > 1 or os.environ.get("PARTMANIP_FORCE_SYNC") == "1":
        return GradSync(name=name)
    return None


def resolve_seed(seed_fn):
    """Run `seed_fn()` on rank 0 and hand its (picklable) result to every rank: a run started with `seed: -1` must not
    draw a different seed -- i.e. different initial weights and a different checkpoint directory -- per rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seed_fn()
    box = [seed_fn() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]
