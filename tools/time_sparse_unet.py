"""SparseUNet backbone alone at cfg 5's cloud shape: forward + backward of a B-cloud mini-batch (default 256 clouds of 4096
'depth_sparse' rows -- short enough for the serialising PMC passes of tools/pmc_run.sh) with HIP events.
usage: python tools/time_sparse_unet.py [B]"""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd.algo_utils import ActorCritic
from partmanip_amd.feeder import FeederEnv
from partmanip_amd import ops
DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = 4096
env = FeederEnv(B, {"depth_sparse": 4 * P, "proprio_state": 0}, 10, DEV, seed=7, point_num=P)
x = env.reset()["depth_sparse"]
import os
net = dict(name="SparseUNet", activation="tanh", point_num=P, grid=50, sparse_top=os.environ.get("SPARSE_TOP", "1") == "1")
torch.manual_seed(0)
ac = ActorCritic(4 * P, 10, dict(action_std=0.1, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
ac.flat()
dy = torch.randn(B, 10, device=DEV)
def run(n):
    for _ in range(n):
        ac.actor.hip_forward(x)
        ac.actor.hip_backward(dy)
run(2)
ops.TIMER.enable("sparse_unet_fwd", "sparse_unet_bwd")
run(3)
f, b = ops.TIMER.mean_ms("sparse_unet_fwd")[0], ops.TIMER.mean_ms("sparse_unet_bwd")[0]
print(f"B={B}: fwd {f:.2f} ms  bwd {b:.2f} ms")
