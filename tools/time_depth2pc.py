"""Time TSDFVolume.depth2pc at the size the reference's comment quotes (depth2tsdf.py:158: "~0.5s for [64, 6, 180, 320]")."""
import sys, time, torch
sys.path.insert(0, '.')
from partmanip_amd.depth2tsdf import TSDFVolume
DEV = 'cuda:0'
b, m, h, w = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 6, 180, 320)))
vol = TSDFVolume(DEV)
pose = torch.eye(4).repeat(m, 1, 1)
for i in range(m):
    pose[i, :3, 3] = torch.tensor([0.02 * i, -0.01 * i, -0.6])
intr = [[250.0, 0.0, w / 2 - 0.5], [0.0, 250.0, h / 2 - 0.5], [0.0, 0.0, 1.0]]
vol.register_camera(pose.numpy(), intr, h, w, b)
torch.manual_seed(0)
depth = torch.rand(b, m, h, w, device=DEV) * 0.5 + 0.45
pc = vol.depth2pc(depth)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    pc = vol.depth2pc(depth)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
valid = float((pc.abs().sum(-1) > 0).float().mean())
print(f"depth2pc [{b},{m},{h},{w}] -> {tuple(pc.shape)}: {dt * 1e3:.1f} ms  ({b / dt:.0f} env/s), non-origin samples {valid:.2f}")
