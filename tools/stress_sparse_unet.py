"""Repeat the cfg 5 forward + backward (fused and materialised gathers) on one 256-cloud batch and compare every repetition with the
first one bit for bit: any difference is a race (all kernels of this path are deterministic by construction).
usage: python tools/stress_sparse_unet.py [reps] [B]"""
import sys, torch
sys.path.insert(0, '.')
from tests.test_gpu_sparse_unet import NET_FULL, _full_size_clouds, _model
from tests.golden import cases
from tests.helpers import t
from partmanip_amd.algo_utils import ActorCritic
from partmanip_amd.autograd import backbone_apply
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
P, A = NET_FULL["point_num"], 10
sd = cases.actor_critic_state(NET_FULL, 4 * P, A, 0.5, 47)
x = _full_size_clouds(B, 771)
w_all = torch.randn(B, A, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
bad = 0
for fused in (True, False):
    ac = ActorCritic(4 * P, A, _model(dict(NET_FULL, fused_gather=fused))).to(DEV)
    ac.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    ac.flat()
    first = None
    for r in range(reps):
        for p in ac.actor.parameters():
            p.grad = None
        out = backbone_apply(ac.actor, x)
        (out * w_all).sum().backward()
        cur = {"out": out.detach().clone(), **{n: p.grad.clone() for n, p in ac.actor.named_parameters()}}
        if first is None:
            first = cur
            continue
        for n in cur:
            if not torch.equal(cur[n], first[n]):
                d = float((cur[n] - first[n]).abs().max()) / max(1.0, float(first[n].abs().max()))
                print(f"fused={fused} rep {r}: {n} differs from rep 0 by {d:.3e} ({int((cur[n] != first[n]).sum())} elements)")
                bad += 1
    del ac
    torch.cuda.empty_cache()
print("repetitions that differed:", bad)
