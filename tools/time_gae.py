"""GAE scan alone: 200 back-to-back launches between two HIP events (4096 envs x 128 steps: 9.4 MB algorithmic)."""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
DEV = 'cuda:0'
for T, N in [(128, 4096), (8, 4096), (128, 32768)]:
    r, v = torch.randn(T, N, 1, device=DEV), torch.randn(T, N, 1, device=DEV)
    d = torch.rand(T, N, 1, device=DEV) < 0.02
    s = d & (torch.rand(T, N, 1, device=DEV) < 0.5)
    last = torch.randn(N, 1, device=DEV)
    ret, adv = torch.empty_like(r), torch.empty_like(r)
    for _ in range(5):
        ops.gae_scan(r, v, d, s, last, ret, adv, 0.99, 0.95, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.gae_scan(r, v, d, s, last, ret, adv, 0.99, 0.95, None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print(f"T={T} N={N}: {us:.1f} us per launch, {T * N * 18 / us / 1e3:.0f} GB/s of 18 B per env-step")
