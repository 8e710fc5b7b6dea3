"""Time the Linear kernels (K4/K5) at the state-PPO shapes, next to torch.mm (hipBLASLt) as a yardstick."""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
DEV = 'cuda:0'
ws = ops.Workspace(DEV)


def t(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M, K, N in ((2048, 512, 512), (2048, 53, 512), (2048, 512, 10), (131072, 128, 128), (131072, 256, 512)):
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    y = torch.empty(M, N, device=DEV)
    dy = torch.randn(M, N, device=DEV)
    dx = torch.empty(M, K, device=DEV)
    dw = torch.empty(N, K, device=DEV)
    db = torch.empty(N, device=DEV)
    fl = 2.0 * M * N * K
    f = t(lambda: ops.linear_fwd(x, w, b, y, ops.ACT_TANH))
    d = t(lambda: ops.linear_bwd_data(dy, w, x, dx, ops.ACT_TANH))
    g = t(lambda: ops.linear_bwd_weight(dy, x, dw, db, ws))
    tm = t(lambda: torch.mm(x, w.t(), out=y))
    print(f"M={M} K={K} N={N}: fwd {f:.1f} us ({fl / f / 1e6:.1f} TF)  dgrad {d:.1f} us  wgrad {g:.1f} us  | torch.mm {tm:.1f} us ({fl / tm / 1e6:.1f} TF)")
