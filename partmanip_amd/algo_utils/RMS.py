"""Running observation statistics with the reference's interface
(algorithms/algo_utils/RMS.py:3-56: `Normalization(shape, device)(x, update)`,
`running_ms.{mean,std,S,n}`, `save()/load()`).  Rollout side and tiny -- tensor plumbing,
outside the kernel scope."""
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

    def update(self, x):
        self.n += 1
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
        return (x - rms.mean) / rms.std


class AdvScaling:
    """RMS.py:48-56 (unused by the runners; divides by the running std only)."""

    def __init__(self, shape, device):
        self.running_ms = RunningMeanStd(shape=shape, device=device)

    def __call__(self, x):
        self.running_ms.update(x.reshape(-1, 1))
        return (x / (self.running_ms.std + 1e-8)).reshape(x.shape[0], x.shape[1], 1)
