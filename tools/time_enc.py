"""Time the fused encoder forward/backward kernels alone (B=2048 clouds) with HIP events."""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd.algo_utils import ActorCritic
DEV = 'cuda:0'
import os
net = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False, precision=os.environ.get("PN_PRECISION", "f32"),
           precision_bwd=os.environ.get("PN_PRECISION_BWD", "f32"), save_h2=os.environ.get("PN_SAVE_H2", "1") == "1")
torch.manual_seed(0)
ac = ActorCritic(3072, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
ac.flat()
B = 2048
x = (torch.rand(B, 1024, 3, device=DEV) * 2 - 1).reshape(B, -1).contiguous()
dy = torch.randn(B, 10, device=DEV)
from partmanip_amd import ops
def run(n):
    for _ in range(n):
        ac.actor.hip_forward(x)
        ac.actor.hip_backward(dy)
run(2)
ops.TIMER.enable("pointnet_enc_fwd", "pointnet_enc_bwd")
run(10)
f = ops.TIMER.mean_ms("pointnet_enc_fwd")[0]; b = ops.TIMER.mean_ms("pointnet_enc_bwd")[0]
print(f"fwd {f:.3f} ms ({688.8/f:.1f} TF)  bwd {b:.3f} ms")
