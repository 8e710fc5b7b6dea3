"""Bit-reproducibility of the PointNet++ backbone ("1000 repetitions of a PointNet2 forward + backward bit-identical").
Since round 6 the level-1 gradient of the per-source-point layer-1 rows is summed in a fixed order over the plan's inverse table
(no floating-point atomics), so every bit of a forward + backward must repeat; `--atomic` runs the old scatter for contrast (its
dY, hence sa.0's gradients, differ in the last bits from run to run).

Each repetition: forward + backward of the default `PointNet2` network (bench `vision_pn2` shapes) on the same clouds, weights and
output gradient, geometry tables + plans cached as the learner caches them; the output and every parameter gradient are hashed
(two 64-bit words each) and compared with repetition 0.  `--noise`: a second stream keeps the chip busy (perturbs the timing).
usage: python tools/stress_pointnet2.py --reps 1000 [--B 512] [--noise] [--atomic]"""
import argparse
import sys
import threading
import time

import torch

sys.path.insert(0, '.')
from tools.stress_sparse_unet import _hash, _noise_loop                            # noqa: E402
from tests.golden import cases                                                      # noqa: E402
from tests.helpers import t                                                         # noqa: E402
from partmanip_amd.algo_utils import ActorCritic                                    # noqa: E402
from partmanip_amd.autograd import backbone_apply                                   # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--B", type=int, default=512)
    ap.add_argument("--noise", action="store_true")
    ap.add_argument("--atomic", action="store_true")
    args = ap.parse_args()
    print(f"stress_pointnet2: {vars(args)}")
    P, A, B = 1024, 10, args.B
    net = dict(name="PointNet2", activation="tanh", sa_deterministic=not args.atomic)
    model = dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)
    sd = cases.actor_critic_state(net, 3 * P, A, 0.5, 47)
    g = torch.Generator(device=DEV).manual_seed(11)
    x = (torch.rand(B, P, 3, device=DEV, generator=g) * 2 - 1 + (torch.rand(B, 1, 3, device=DEV, generator=g) - 0.5)).reshape(B, 3 * P)
    w_all = torch.randn(B, A, device=DEV, generator=g)
    ac = ActorCritic(3 * P, A, model).to(DEV)
    ac.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    ac.flat()
    tabs = ac.actor.precompute_geometry(x)
    ac.actor.precompute_plans(tabs, x, [(0, B)])
    stop, th = [False], None
    if args.noise:
        th = threading.Thread(target=_noise_loop, args=(stop, torch.cuda.Stream()), daemon=True)
        th.start()
    names = [n for n, _ in ac.actor.named_parameters()] + ["out"]
    first, bad, t0 = None, 0, time.time()
    for r in range(args.reps):
        for p in ac.actor.parameters():
            p.grad = None
        ac.actor.use_geometry(tabs, (0, B))
        out = backbone_apply(ac.actor, x)
        (out * w_all).sum().backward()
        cur = torch.stack([_hash(p.grad) for p in ac.actor.parameters()] + [_hash(out)]).cpu()
        if first is None:
            first = cur
        elif not torch.equal(cur, first):
            bad += 1
            if bad <= 5:
                print(f"rep {r}: differs from rep 0 in", [names[i] for i in (cur != first).any(dim=1).nonzero().view(-1).tolist()])
    stop[0] = True
    if th is not None:
        th.join()
    print(f"PointNet2 forward + backward x {args.reps} at {B} clouds ({'atomic scatter' if args.atomic else 'fixed-order sums'}"
          f"{', noise stream' if args.noise else ''}) in {time.time() - t0:.1f} s")
    print("repetitions that differed:", bad)
    return 1 if (bad and not args.atomic) else 0


if __name__ == "__main__":
    sys.exit(main())
