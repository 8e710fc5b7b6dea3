import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import ref_cpu as R
from partmanip_amd.algo_utils import ActorCritic
DEV='cuda:0'
net = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False)
torch.manual_seed(257)
ac = ActorCritic(3072, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), 0).to(DEV)
f = ac.flat()
g = torch.Generator().manual_seed(257)
B=257
x = (torch.rand(B, 1024, 3, generator=g)*2-1).reshape(B, -1).contiguous()
dy = torch.randn(B, 10, generator=g)
def run(xs, dys):
    ac.actor.hip_forward(xs.to(DEV)); ac.actor.hip_backward(dys.to(DEV)); torch.cuda.synchronize()
    return f["grad_actor"][:f["n_actor"]].clone()
g257 = run(x, dy)
g256 = run(x[:256].contiguous(), dy[:256].contiguous())
g1 = run(x[256:].contiguous(), dy[256:].contiguous())      # cloud 256 alone (B=1)
d = g257 - g256
print("cloud256 via diff vs alone: max|diff-alone|", float((d-g1).abs().max()), "max|alone|", float(g1.abs().max()), "max|d|", float(d.abs().max()))
# swap order: put cloud 256 first
perm = torch.cat([torch.tensor([256]), torch.arange(256)])
g257p = run(x[perm].contiguous(), dy[perm].contiguous())
print("perm invariance max diff", float((g257p-g257).abs().max()), "scale", float(g257.abs().max()))
off=0
for k,v in ac.actor.named_parameters():
    n=v.numel(); print(k, float((d-g1)[off:off+n].abs().max()), float(g1[off:off+n].abs().max())); off+=n
