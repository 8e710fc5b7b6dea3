"""Time pm_fps_f32 at the learner's sizes (2048 clouds: 1024 -> 256 centres, 256 -> 64 centres)."""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
DEV = 'cuda:0'
ws = ops.Workspace(DEV)
for B, P, K in ((2048, 1024, 256), (2048, 256, 64), (4096, 1024, 256), (256, 2048, 512), (64, 8192, 1024)):
    xyz = torch.rand(B, P, 3, device=DEV) * 2 - 1
    ops.fps(xyz, K, ws)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.fps(xyz, K, ws)
    b.record()
    torch.cuda.synchronize()
    print(f"fps B={B} P={P} K={K}: {a.elapsed_time(b) / 5:.3f} ms")
