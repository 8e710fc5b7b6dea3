"""Time the rollout-side kernels at the benchmark's observation size: Normalization over a
(4096, 3072) batch (pm_rms_update_f32 + pm_rms_normalize_f32) next to the reference's tensor expression on the same
device, and the sampling tail of random_act_cri (pm_gaussian_sample_f32) next to its torch ops."""
import sys, time, torch
sys.path.insert(0, '.')
from partmanip_amd.algo_utils import Normalization
from partmanip_amd import ops
DEV = 'cuda:0'
N, D, A = 4096, 3072, 10
x = torch.randn(N, D, device=DEV) * 2 + 1


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


norm = Normalization(D, DEV)


class TorchRMS:                                       # RMS.py:10-18,40-45 verbatim semantics, torch ops on the GPU
    def __init__(self):
        self.n, self.mean, self.S = 0, torch.zeros(1, D, device=DEV), torch.full((1, D), 1e-4, device=DEV)

    def __call__(self, x):
        self.n += 1
        old = self.mean.clone()
        new = x.mean(dim=0, keepdim=True)
        self.mean = old + (new - old) / self.n
        self.S = self.S + (x - new).pow(2).mean(dim=0, keepdim=True) + (old - new).pow(2) * (self.n - 1) / self.n
        self.std = torch.sqrt(self.S / self.n)
        return (x - self.mean) / self.std


tr = TorchRMS()
a, b = timed(lambda: norm(x)), timed(lambda: tr(x))
gb = N * D * 4 * 3 / 1e9                              # algorithmic: read for the moments, read + write to normalise
print(f"Normalization ({N}, {D}): HIP {a:.0f} us ({gb / a * 1e6:.0f} GB/s of 3 passes)   torch expression {b:.0f} us")
mu = torch.randn(N, A, device=DEV)
ls = torch.full((A,), -0.7, device=DEV)
eps = torch.randn(N, A, device=DEV)


def torch_tail():
    s2 = ls.exp() * ls.exp()
    xx = mu + s2 * eps
    z = (xx - mu) / s2
    lp = -0.5 * (z * z).sum(-1) - torch.log(s2).sum() - 0.5 * A * 1.8378770664093453
    return torch.tanh(xx), lp, ls.repeat(N, 1)


a, b = timed(lambda: ops.gaussian_sample(mu, ls, eps, 1.0, True)), timed(torch_tail)
print(f"random_act_cri tail ({N}, {A}): HIP {a:.0f} us   torch ops {b:.0f} us")
