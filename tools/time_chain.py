"""Hidden-layer chain of the small-step regime (2048 x 53 -> 512 -> 512 -> 512, two data gradients) as ONE launch each way
(pm_linear_fwd_chain_f32 / pm_linear_bwd_data_chain_f32) against the layer-by-layer launches, 32 repetitions replayed from a
hipGraph on the otherwise idle chip: us per forward chain / per data-gradient chain."""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
DEV = torch.device('cuda:0')
M, O, H = 2048, 53, 512
torch.manual_seed(0)
x = torch.randn(M, O, device=DEV)
Ws = [torch.randn(H, O, device=DEV) / O ** 0.5, torch.randn(H, H, device=DEV) / H ** 0.5, torch.randn(H, H, device=DEV) / H ** 0.5]
bs = [torch.zeros(H, device=DEV) for _ in range(3)]
ys = [torch.empty(M, H, device=DEV) for _ in range(3)]
dz = torch.randn(M, H, device=DEV)
dxs = [torch.empty(M, H, device=DEV) for _ in range(2)]
ws = ops.Workspace(DEV)
A = ops.ACT_TANH
def fwd_layers():
    cur = x
    for W, b, y in zip(Ws, bs, ys):
        ops.linear_fwd(cur, W, b, y, A); cur = y
def fwd_chain():
    ins = [x] + ys[:-1]
    assert ops.linear_fwd_chain([(i, W, b, y, A) for i, W, b, y in zip(ins, Ws, bs, ys)], ws)
def bwd_layers():
    ops.linear_bwd_data(dz, Ws[2], ys[1], dxs[0], A); ops.linear_bwd_data(dxs[0], Ws[1], ys[0], dxs[1], A)
def bwd_chain():
    assert ops.linear_bwd_data_chain([(dz, Ws[2], ys[1], dxs[0], A), (dxs[0], Ws[1], ys[0], dxs[1], A)], ws)
def timed(fn, n=32):
    fn()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, fn in (("fwd 3 layers, 3 launches", fwd_layers), ("fwd 3 layers, 1 chained launch", fwd_chain),
                 ("dgrad 2 layers, 2 launches", bwd_layers), ("dgrad 2 layers, 1 chained launch", bwd_chain)):
    print(f"{name:36s} {timed(fn):7.1f} us")
print("gave up:", ops.chain_gave_up(ws))
