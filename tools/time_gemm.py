"""Time the Linear kernels (K4/K5) at the state-PPO and PointNet++-glue shapes, next to torch.mm (hipBLASLt) as a yardstick.
Launches are captured into a hipGraph and replayed: an eager Python/ctypes launch loop is host-bound below ~15 us per call
and reads the same ~18 us for every kernel at the 2048-row shapes."""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
DEV = 'cuda:0'
ws = ops.Workspace(DEV)
REP = 20


def t(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
    torch.cuda.synchronize()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * REP) * 1e3


shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or \
    [(2048, 512, 512), (2048, 53, 512), (2048, 512, 10), (131072, 128, 128), (131072, 256, 512)]
for M, K, N in shapes:
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    y = torch.empty(M, N, device=DEV)
    dy = torch.randn(M, N, device=DEV)
    dx = torch.empty(M, K, device=DEV)
    dw = torch.empty(N, K, device=DEV)
    db = torch.empty(N, device=DEV)
    fl = 2.0 * M * N * K
    ws.get(ops.lib.pm_linear_bwd_weight_workspace_bytes(M, N, K))
    f = t(lambda: ops.linear_fwd(x, w, b, y, ops.ACT_TANH))
    d = t(lambda: ops.linear_bwd_data(dy, w, x, dx, ops.ACT_TANH))
    g = t(lambda: ops.linear_bwd_weight(dy, x, dw, db, ws))
    tm = t(lambda: torch.mm(x, w.t(), out=y))
    print(f"M={M} K={K} N={N}: fwd {f:.1f} us ({fl / f / 1e6:.1f} TF)  dgrad {d:.1f} us ({fl / d / 1e6:.1f} TF)  "
          f"wgrad {g:.1f} us ({fl / g / 1e6:.1f} TF)  | torch.mm {tm:.1f} us ({fl / tm / 1e6:.1f} TF)")
