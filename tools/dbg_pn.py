import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import ref_cpu as R
from partmanip_amd.algo_utils import ActorCritic
DEV='cuda:0'
def rel(got, ref):
    ref = ref.double(); return float((got.double().cpu()-ref).abs().max()/(ref.abs().max()+1e-30))
net = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False)
for B in (6, 64, 256, 257, 300):
    torch.manual_seed(B)
    ac = ActorCritic(3072, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), 0).to(DEV)
    f = ac.flat()
    g = torch.Generator().manual_seed(B)
    x = (torch.rand(B, 1024, 3, generator=g)*2-1).reshape(B, -1).contiguous()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ac.state_dict().items()}
    out_ref = R.pointnet_forward(p, "actor", net, x.clone(), 0)
    dy = torch.randn(B, 10, generator=g)
    names = [k for k in p if k.startswith("actor.")]
    grads_ref = torch.autograd.grad((out_ref*dy).sum(), [p[k] for k in names])
    out = ac.actor.hip_forward(x.to(DEV))
    ac.actor.hip_backward(dy.to(DEV))
    torch.cuda.synchronize()
    off = 0; errs = {}
    for k, v in ac.actor.named_parameters():
        errs[k] = rel(f["grad_actor"][off:off+v.numel()].view(v.shape), grads_ref[names.index("actor."+k)]); off += v.numel()
    print(B, "fwd", rel(out, out_ref.detach()), {k: f"{e:.1e}" for k, e in errs.items()})
    # determinism: run backward again
    g1 = f["grad_actor"].clone()
    ac.actor.hip_forward(x.to(DEV)); ac.actor.hip_backward(dy.to(DEV)); torch.cuda.synchronize()
    print("   rerun max diff", float((g1 - f["grad_actor"]).abs().max()))
