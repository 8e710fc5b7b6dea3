"""The two big submanifold convolutions of cfg 5 as bare kernels: the gathered GEMM (forward / weight gradient, neighbour rows
fetched inside the LDS-DMA loader) next to the SAME GEMM on a materialised operand (what the gather would cost if it were free
is the difference) -- rows x (27 C) x C at the level sizes of a 2048-cloud mini-batch.  ms and TFLOP/s per launch."""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
DEV = torch.device('cuda:0')
ws = ops.Workspace(DEV)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rows, C in ((2684518, 64), (624221, 128)):
    J = 27
    g = torch.Generator(device=DEV).manual_seed(1)
    src = torch.randn(rows, C, device=DEV, generator=g)
    # neighbours like a surface in cell order: mostly nearby rows, ~45 % absent
    base = torch.arange(rows, device=DEV).view(-1, 1)
    off = torch.randint(-40, 41, (rows, J), device=DEV, generator=g)
    idx = (base + off).clamp_(0, rows - 1)
    idx = torch.where(torch.rand(rows, J, device=DEV, generator=g) < 0.45, torch.full_like(idx, -1), idx).to(torch.int32).contiguous()
    w = torch.randn(C, J * C, device=DEV, generator=g) / (J * C) ** 0.5
    b = torch.zeros(C, device=DEV)
    y = torch.empty(rows, C, device=DEV)
    dz = torch.randn(rows, C, device=DEV, generator=g)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    zero = torch.zeros(C + 64, device=DEV)
    fl = 2.0 * rows * J * C * C
    t = timed(lambda: ops.sparse_conv_fwd(src, idx, C, w, b, y, ops.ACT_TANH, zero))
    print(f"rows {rows} C {C}: gathered fwd   {t:6.2f} ms {fl / t / 1e9:6.1f} TF")
    t = timed(lambda: ops.sparse_conv_bwd_weight(dz, src, idx, C, dw, db, zero, ws))
    print(f"rows {rows} C {C}: gathered wgrad {t:6.2f} ms {fl / t / 1e9:6.1f} TF")
    if rows * J * C * 4 < 40e9:
        cols = torch.empty(rows, J * C, device=DEV)
        ops.rows_gather(src, idx, C, cols)
        t = timed(lambda: ops.linear_fwd(cols, w, b, y, ops.ACT_TANH))
        print(f"rows {rows} C {C}: plain fwd      {t:6.2f} ms {fl / t / 1e9:6.1f} TF   (operand materialised: {cols.numel() * 4 / 1e9:.1f} GB)")
        t = timed(lambda: ops.linear_bwd_weight(dz, cols, dw, db, ws))
        print(f"rows {rows} C {C}: plain wgrad    {t:6.2f} ms {fl / t / 1e9:6.1f} TF")
        t = timed(lambda: torch.mm(cols, w.t(), out=y))
        print(f"rows {rows} C {C}: torch.mm fwd   {t:6.2f} ms {fl / t / 1e9:6.1f} TF")
        del cols
