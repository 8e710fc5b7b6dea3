"""Running observation statistics with the reference's interface
(algorithms/algo_utils/RMS.py:3-56: `Normalization(shape, device)(x, update)`,
`running_ms.{mean,std,S,n}`, `save()/load()`).  Rollout side: on a GPU
device the batch moments and the normalisation run in pm_rms_update_f32 / pm_rms_normalize_f32
(one read of the batch for the update, one read + one write to normalise); a CPU device keeps
the reference's tensor expressions (checkpoint tooling, CPU tests of the host logic)."""
import torch

_KEYS = ("mean", "std", "S", "n")


class RunningMeanStd:
    """Batch-wise running mean / std: each `update` folds in one (N, shape) batch with weight 1
    (RMS.py:10-18 -- note `n` counts batches, not samples)."""

    def __init__(self, shape, device):
        self.n = 0
        self.mean = torch.zeros((1, shape), device=device)
        self.S = torch.full((1, shape), 1e-4, device=device)
        self.std = self.S.sqrt()
        # data parallelism (partmanip_amd.dist.GradSync, set by the runner): every rank sees an env shard of the batch;
        # the column moments are summed over ranks so that all replicas keep the statistics of the WHOLE batch
        self.sync = None

    def _hip(self, x):
        return x.is_cuda and x.dim() == 2 and x.dtype == torch.float32

    def update(self, x):
        self.n += 1
        if self._hip(x):
            from .. import ops
            if getattr(self, "_ws", None) is None:
                self._ws = ops.Workspace(x.device)
            # state tensors may have been replaced by load(): keep them fp32, contiguous and on the batch's device
            for k in ("mean", "S", "std"):
                t = getattr(self, k)
                if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                    setattr(self, k, t.to(device=x.device, dtype=torch.float32).contiguous())
            sync = getattr(self, "sync", None)
            if sync is not None:
                if getattr(self, "_mom", None) is None or self._mom.numel() != 2 * x.shape[1]:
                    self._mom = torch.empty(2 * x.shape[1], dtype=torch.float64, device=x.device)
                ops.rms_update_dp(x, self.n, self.mean, self.S, self.std, self._ws, self._mom, sync.sum_, sync.world)
                return
            ops.rms_update(x, self.n, self.mean, self.S, self.std, self._ws)
            return
        if getattr(self, "sync", None) is not None:
            raise RuntimeError("data-parallel RunningMeanStd needs device batches (the moments are all-reduced on the GPU)")
        prev = self.mean.clone()
        cur = x.mean(dim=0, keepdim=True)
        within = (x - cur).pow(2).mean(dim=0, keepdim=True)
        between = (prev - cur).pow(2) * (self.n - 1) / self.n
        self.mean = prev + (cur - prev) / self.n
        self.S = self.S + within + between
        self.std = torch.sqrt(self.S / self.n)

    def save(self):
        return {k: getattr(self, k) for k in _KEYS}

    def load(self, load_dict):
        for k in _KEYS:
            setattr(self, k, load_dict[k])


class Normalization:
    def __init__(self, shape, device):
        self.running_ms = RunningMeanStd(shape=shape, device=device)

    def __call__(self, x, update=True):
        rms = self.running_ms
        if update:
            rms.update(x)
        if rms._hip(x) and rms.mean.is_cuda:
            from .. import ops
            return ops.rms_normalize(x, rms.mean.contiguous(), rms.std.contiguous())
        return (x - rms.mean) / rms.std


class AdvScaling:
    """RMS.py:48-56 (unused by the runners; divides by the running std only)."""

    def __init__(self, shape, device):
        self.running_ms = RunningMeanStd(shape=shape, device=device)

    def __call__(self, x):
        self.running_ms.update(x.reshape(-1, 1))
        return (x / (self.running_ms.std + 1e-8)).reshape(x.shape[0], x.shape[1], 1)
