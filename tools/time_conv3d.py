"""Conv3DNet's input layer alone at the shipped DAgger geometry (1600 volumes of 50^3, k 5, stride 3, 16 filters): forward and
weight gradient, 20 launches each between HIP events.  PARTMANIP_HIP_LIB selects an A/B build."""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
DEV = 'cuda:0'
B, r = 1600, 50
ring = torch.randn(4000, r ** 3 + 8, device=DEV)
rows = torch.randperm(4000, device=DEV)[:B]
x5 = ring[:, :r ** 3].unflatten(1, (1, r, r, r))
wt = torch.randn(125, 16, device=DEV) * 0.1
b = torch.randn(16, device=DEV) * 0.1
ws = ops.Workspace(DEV)
y = ops.conv3d_c1_fwd(x5, 5, 3, 2, wt, b, ops.ACT_TANH, rows)
dz = torch.randn_like(y)
dw, db = torch.empty(16, 125, device=DEV), torch.empty(16, device=DEV)
def t(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
f = t(lambda: ops.conv3d_c1_fwd(x5, 5, 3, 2, wt, b, ops.ACT_TANH, rows))
g = t(lambda: ops.conv3d_c1_wgrad(dz, x5, 5, 3, 2, dw, db, ws, rows))
nb = B * r ** 3 * 4 + y.numel() * 4
print(f"conv1 fwd {f:.3f} ms ({nb / f / 1e6:.0f} GB/s)   wgrad {g:.3f} ms ({nb / g / 1e6:.0f} GB/s)")
