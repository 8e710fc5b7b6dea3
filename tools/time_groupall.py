#!/usr/bin/env python3
"""PointNet++ group-all level at the bench size (2048 clouds x 64 rows, 256 -> 512): the fused kernels of csrc/sa_groupall.hip
against the Linear + max-pool launches they replace (forward; backward = pooled gradient -> dH, dW, db).

    python tools/time_groupall.py [clouds]
"""
import sys
import time

import torch

sys.path.insert(0, '.')
from partmanip_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
R, CK, CO = 64, 256, 512
dev = torch.device("cuda:0")
torch.manual_seed(0)
h = torch.tanh(torch.randn(B * R, CK, device=dev))
W = torch.randn(CO, CK, device=dev) / 16
b = torch.randn(CO, device=dev) * 0.1
dfeat = torch.randn(B, CO, device=dev)
ws = ops.Workspace(dev)
packed = torch.empty(int(ops.lib.pm_sa_groupall_packed_elems(CK, CO)), device=dev)
feat = torch.empty(B, CO, device=dev)
dh, dW, db = torch.empty_like(h), torch.empty_like(W), torch.empty_like(b)
y = torch.empty(B * R, CO, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fwd_fused():
    ops.sa_groupall_pack(W, packed)
    return ops.sa_groupall_fwd(h, B, R, b, packed, feat)


def fwd_plain():
    ops.linear_fwd(h, W, b, y, ops.ACT_TANH)
    return ops.maxpool_rows(y, B, R, feat)


arg = fwd_fused()


def bwd_fused():
    ops.sa_groupall_bwd(dfeat, feat, arg, W, h, B, R, dh, dW, db, ws)


arg2 = fwd_plain()


def bwd_plain():
    dy = ops.maxpool_rows_bwd(dfeat, arg2, R, y_tanh=y)
    ops.linear_bwd_weight(dy, h, dW, db, ws)
    ops.linear_bwd_data(dy, W, h, dh, ops.ACT_TANH)


flops = 2.0 * B * R * CK * CO
tf, tp = timed(fwd_fused), timed(fwd_plain)
bf, bp = timed(bwd_fused), timed(bwd_plain)
print(f"clouds {B}: forward fused {tf:.3f} ms ({flops / tf / 1e9:.1f} TFLOP/s) vs Linear + max-pool {tp:.3f} ms; "
      f"backward fused {bf:.3f} ms vs scatter + two GEMMs {bp:.3f} ms")
