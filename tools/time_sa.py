"""Time the fused set-abstraction kernels alone (B=2048 clouds, default PointNet2 levels) with HIP events.
usage: python tools/time_sa.py [B]   (env PARTMANIP_HIP_LIB selects an A/B build of the library)"""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
ws = ops.Workspace(DEV)
xyz = torch.rand(B, 1024, 3, device=DEV) * 2 - 1
LEVELS = [("A", 1024, 256, 0.2, (64, 64, 128), 0), ("B", 256, 64, 0.4, (128, 128, 256), 128)]
feat = None
for name, P, S, radius, dims, cf in LEVELS:
    C1, C2, C3 = dims
    idx_c = ops.fps(xyz, S, ws)
    centers = ops.group_points(xyz, idx_c.view(B, S, 1)).view(B, S, 3)
    idx_g = ops.ball_query(xyz, centers, radius, 32)
    ldw1 = (3 + cf + 3) // 4 * 4
    W1 = torch.randn(C1, ldw1, device=DEV) * 0.3
    W2 = torch.randn(C2, C1, device=DEV) / C1 ** 0.5
    W3 = torch.randn(C3, C2, device=DEV) / C2 ** 0.5
    b1, b2, b3 = (torch.randn(c, device=DEV) * 0.1 for c in dims)
    packed = torch.empty(int(ops.lib.pm_sa_packed_elems(*dims)), device=DEV)
    ops.sa_pack(W2, W3, packed)
    Y = None
    if cf:
        feat = torch.randn(B * P, cf, device=DEV) * 0.5
        Y = torch.empty(B * P, C1, device=DEV)
        ops.linear_fwd(feat, W1[:, 3:3 + cf], None, Y, ops.ACT_NONE)
    pooled = torch.empty(B * S, C3, device=DEV)
    dpooled = torch.randn(B * S, C3, device=DEV)
    g = [torch.empty_like(t) for t in (W1, b1, W2, b2, W3, b3)]
    dY = torch.zeros(B * P, C1, device=DEV) if cf else None
    import os
    h2 = torch.empty(B * S * 32, C2, device=DEV) if os.environ.get("SA_SAVE_H2", "1") == "1" else None

    PACKED = os.environ.get("SA_PACKED", "1") == "1"       # duplicate-free rows (default) or the dense 32-row groups
    DET = os.environ.get("SA_DET", "1") == "1" and cf > 0     # dz1 rows + fixed-order sums (default) or fp32 atomics into dY
    plan = ops.sa_plan(idx_g, xyz, centers, dims, ws, inverse=DET) if PACKED else None
    dz1 = torch.empty(plan.counts()[0], C1, device=DEV) if (PACKED and DET) else None
    if PACKED and DET:
        pw = torch.empty(int(ops.lib.pm_sa_dy_consume_packed_elems(C1, cf)), device=DEV)
        ops.sa_dy_consume_pack(W1, cf, pw)
        dfeat, dW1c, dW1f = torch.empty(B * P, cf, device=DEV), torch.empty_like(W1), torch.empty(C1, cf, device=DEV)
        w1f = W1[:, 3:3 + cf].contiguous()

    def run(n):
        for _ in range(n):
            if PACKED:
                pl = ops.sa_plan(idx_g, xyz, centers, dims, ws) if os.environ.get('SA_PLAN_EACH', '0') == '1' else plan
                arg = ops.sa_fwd_packed(pl, Y, W1, b1, b2, b3, packed, dims, pooled, h2)
                if DET:
                    ops.sa_bwd_packed(pl, Y, W1, b1, b2, W3, packed, dims, pooled, arg, dpooled, *g, None, ws, h2, dz1=dz1)
                    if os.environ.get("SA_FUSED_DY", "1") == "1":
                        ops.sa_dy_consume(pl, dz1, feat, pw, dfeat, dW1c, ws)
                    else:
                        ops.sa_dy_segsum(pl, dz1, dY)
                        with ops.TIMER.bracket("dy_gemms"):
                            ops.linear_bwd_weight(dY, feat, dW1f, None, ws)
                            ops.linear_bwd_data(dY, w1f, None, dfeat, ops.ACT_NONE)
                else:
                    if dY is not None:
                        ops.fill_zero(dY) if hasattr(ops, "fill_zero") else dY.zero_()
                    ops.sa_bwd_packed(pl, Y, W1, b1, b2, W3, packed, dims, pooled, arg, dpooled, *g, dY, ws, h2)
            else:
                arg = ops.sa_fwd(xyz, centers, idx_g, Y, W1, b1, b2, b3, packed, dims, pooled, h2)
                ops.sa_bwd(xyz, centers, idx_g, Y, W1, b1, b2, W3, packed, dims, pooled, arg, dpooled, *g, dY, ws, h2)
    run(2)
    nf, nb = f"sa_fwd_{C1}x{C2}x{C3}", f"sa_bwd_{C1}x{C2}x{C3}"
    ops.TIMER.enable(nf, nb, "sa_plan", "sa_dy_segsum", "sa_dy_consume", "dy_gemms")
    run(5)
    f = ops.TIMER.mean_ms(nf)[0]
    b = ops.TIMER.mean_ms(nb)[0]
    pt = (ops.TIMER.mean_ms("sa_plan") or [0.0])[0] if PACKED else 0.0
    ops.TIMER.disable()
    rows = B * S * 32
    if PACKED:
        R_, T_ = plan.counts()
        print(f"level {name}: plan {pt:.3f} ms, {R_} distinct rows of {rows} ({R_ / (B * S):.2f} per group), {T_} tiles "
              f"({R_ / T_:.1f} rows per tile)")
        rows = R_
    ff = 2.0 * rows * (C1 * C2 + C2 * C3) / 1e9                 # MFMA flops fwd
    fb = 2.0 * rows * ((2 if h2 is not None else 3) * C1 * C2 + C2 * C3) / 1e9   # [L2 recompute +] dW2 + dH1 + dH2
    sg = ops.TIMER.mean_ms("sa_dy_segsum") if (PACKED and DET) else None
    print(f"level {name}: fwd {f:.3f} ms ({ff / f:.1f} TF)  bwd {b:.3f} ms ({fb / b:.1f} TF executed)"
          + (f"  dY segsum {sg[0]:.3f} ms" if sg else "")
          + "".join(f"  {n_} {ops.TIMER.mean_ms(n_)[0]:.3f} ms" for n_ in ("sa_dy_consume", "dy_gemms") if ops.TIMER.mean_ms(n_)))
    xyz = centers.contiguous()
