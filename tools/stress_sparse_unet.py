"""Race hunt for the SparseUNet backbone (`down1`'s weight gradient of the fused and the materialised backward
once differed by 5.5e-5 instead of 1.8e-7 in ~45 executions).  Every kernel of this path is deterministic by construction (no
floating-point atomics, fixed-order reductions), so ANY bit that changes between two repetitions of the same forward + backward
is a race, a stale read or an uninitialised pad.

Each repetition runs forward + backward of the cfg 5 network on the same clouds and weights.  Every launch through
`partmanip_amd.ops` that the backbone makes is logged (tensor arguments and results are KEPT until the end of the repetition --
no kernel is inserted between the launches, the timing stays the product's) and hashed at the end of the repetition; the log of
repetition r is compared with repetition 0 entry by entry and the FIRST differing launch is named.

  --mode fused | materialised | both     gathers inside the GEMM loaders (product default) / materialised operands (A/B form)
  --noise                                a second stream keeps the chip busy with unrelated launches (perturbs the timing)
  --perlaunch                            hash right after each launch instead (serialises: the control that hides timing races)
  --dagger                               repeat `dagger.update` (random sampler, 4 mini-batches x 2 epochs) with the two-stream
                                         geometry prefetch on (PARTMANIP_GEOM_PREFETCH=1) / off and compare parameters + loss
usage: python tools/stress_sparse_unet.py --reps 1000 [--B 256] [--mode both] [--noise] [--perlaunch] [--dagger]
(AMD_SERIALIZE_KERNEL=3 in the environment is the other control: every launch waits for the one before.)"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from tests.test_gpu_sparse_unet import NET_FULL, _full_size_clouds, _model          # noqa: E402
from tests.golden import cases                                                      # noqa: E402
from tests.helpers import t, FakeEnv, FakeLogger                                    # noqa: E402
from partmanip_amd import ops                                                       # noqa: E402
from partmanip_amd.algo_utils import ActorCritic                                    # noqa: E402
from partmanip_amd.autograd import backbone_apply                                   # noqa: E402

DEV = "cuda:0"
LOGGED = ("voxel_grid0", "voxel_down", "voxel_nbr27", "voxel_mirror27", "sparse_conv_fwd", "sparse_conv_bwd_data", "sparse_conv_bwd_weight",
          "linear_fwd", "linear_bwd_data", "linear_bwd_weight", "rows_gather", "rows_gather_bwd", "maxpool_rows", "maxpool_rows_bwd")


def _hash(x):
    """Two 64-bit words of a tensor's bits: the plain sum and a position-weighted sum (a permutation changes the second)."""
    v = x.detach().contiguous().view(-1)
    if v.dtype in (torch.float32, torch.int32):
        v = v.view(torch.int32).to(torch.int64)
    elif v.dtype == torch.int64:
        pass
    else:
        v = v.to(torch.int64)
    w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521) + 1
    return torch.stack([v.sum(), (v * w).sum()])


def _tensors(obj, out):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda and obj.numel() > 0:
            out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _tensors(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            _tensors(o, out)


class LaunchLog:
    """Wraps the ops the backbone calls; keeps (name, tensors) per launch; hashes at the end of the repetition or per launch."""

    def __init__(self, perlaunch):
        self.perlaunch, self.entries, self.orig = perlaunch, [], {}

    def __enter__(self):
        for name in LOGGED:
            fn = getattr(ops, name)
            self.orig[name] = fn
            setattr(ops, name, self._wrap(name, fn))
        return self

    def __exit__(self, *a):
        for name, fn in self.orig.items():
            setattr(ops, name, fn)

    def _wrap(self, name, fn):
        def call(*a, **k):
            r = fn(*a, **k)
            ts = []
            _tensors((a, k, r), ts)
            self.entries.append((name, [_hash(x) for x in ts] if self.perlaunch else ts))
            return r
        return call

    def finish(self):
        """-> (names, (n_words,) int64 tensor on the host)"""
        names, words = [], []
        for k, (name, ts) in enumerate(self.entries):
            hs = ts if self.perlaunch else [_hash(x) for x in ts]
            for j, h in enumerate(hs):
                names.append(f"launch {k} {name}[tensor {j}]")
                words.append(h)
        self.entries = []
        return names, torch.stack(words).cpu()


def _noise_loop(stop, stream):
    a = torch.randn(2048, 2048, device=DEV)
    b = torch.empty(64 << 20, device=DEV)
    with torch.cuda.stream(stream):
        while not stop[0]:
            for _ in range(8):
                b.normal_()
                (a @ a).sum()
            stream.synchronize()


def stress_backbone(args):
    import threading
    P, A, B = NET_FULL["point_num"], 10, args.B
    sd = cases.actor_critic_state(NET_FULL, 4 * P, A, 0.5, 47)
    x = _full_size_clouds(B, 771)
    w_all = torch.randn(B, A, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    bad = 0
    modes = {"fused": [True], "materialised": [False], "both": [True, False]}[args.mode]
    stop, th = [False], None
    if args.noise:
        th = threading.Thread(target=_noise_loop, args=(stop, torch.cuda.Stream()), daemon=True)
        th.start()
    for fused in modes:
        ac = ActorCritic(4 * P, A, _model(dict(NET_FULL, fused_gather=fused))).to(DEV)
        ac.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
        ac.flat()
        first, t0 = None, time.time()
        for r in range(args.reps):
            for p in ac.actor.parameters():
                p.grad = None
            with LaunchLog(args.perlaunch) as log:
                out = backbone_apply(ac.actor, x)
                (out * w_all).sum().backward()
                names, words = log.finish()
            grads = torch.stack([_hash(p.grad) for p in ac.actor.parameters()] + [_hash(out)]).cpu()
            gnames = [n for n, _ in ac.actor.named_parameters()] + ["out"]
            if first is None:
                first = (names, words, grads)
                print(f"fused={fused}: {len(set(n.split('[')[0] for n in names))} launches, {len(words)} hashed tensors per repetition")
                continue
            if names != first[0] or not torch.equal(words, first[1]) or not torch.equal(grads, first[2]):
                bad += 1
                if names != first[0]:
                    print(f"fused={fused} rep {r}: a different launch sequence")
                    continue
                diff = (words != first[1]).any(dim=1).nonzero().view(-1).tolist()
                gd = [gnames[i] for i in (grads != first[2]).any(dim=1).nonzero().view(-1).tolist()]
                where = f"first differing logged tensor: #{diff[0]} {names[diff[0]]} (of {len(diff)} differing)" if diff else "no logged tensor differs"
                print(f"fused={fused} rep {r}: {where}; differing results: {gd}")
        print(f"fused={fused}: {args.reps} repetitions in {time.time() - t0:.1f} s, {bad} differed so far")
        del ac
        torch.cuda.empty_cache()
    stop[0] = True
    if th is not None:
        th.join()
    return bad


def stress_dagger(args):
    """`dagger.update` from the same state with the same sampler seed, the geometry prefetch on and off: parameters and loss must be
    bit-identical across repetitions AND across the two settings."""
    import tempfile
    from partmanip_amd.algorithms import ppo, dagger
    from partmanip_amd.feeder import FeederEnv
    P, A, N, buf, O_t = NET_FULL["point_num"], 10, args.B, 4, 53
    O_s = 4 * P
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    tnet = dict(name="MLP", hid_dim=[64, 64], activation="tanh")
    tcfg = dict(num_envs=N, obs_mode="normal_state", succ_value=None, model=_model(tnet, 0.5), max_iterations=10, n_steps=1, n_updates=1,
                n_minibatches=1, device=DEV, eval_round=1, eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False,
                save_video=False, lr_schedule="fixed", lr=1e-3, desired_kl=0.1, epsilon_clip=0.2, gamma=0.99, lam=0.95,
                tricks=dict(cases.TRICKS_DEFAULT), sampler="sequential", resume=None)
    tea = ppo(FakeEnv(N, {"normal_state": O_t}, A), tcfg, FakeLogger(tmp))
    tea.save(1)
    env = FeederEnv(N, {"normal_state": O_t, "depth_sparse": O_s, "proprio_state": 0}, A, DEV, seed=99, point_num=P)
    rings = [(o["depth_sparse"], o["normal_state"]) for o in (env.reset() for _ in range(buf))]
    init = cases.actor_critic_state(NET_FULL, O_s, A, 0.1, 51)
    first, bad = None, 0
    t0 = time.time()
    for r in range(args.reps):
        os.environ["PARTMANIP_GEOM_PREFETCH"] = "1" if r % 2 == 0 else "0"
        cfg = dict(num_envs=N, obs_mode="depth_sparse", model=_model(NET_FULL, 0.1), max_iterations=100, n_steps=1, n_updates=2, n_minibatches=4,
                   device=DEV, buf_size=buf, reward_reset=False, add_proprio_obs=False, offline_data_pth=None, eval_round=1,
                   eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False, lr_schedule="fixed",
                   lr=1e-3, teacher=os.path.join(tmp, "model_1.pth"), resume=None, pretrain=None, sampler="random")
        run = dagger(env, cfg, FakeLogger(tmp))
        run.student.load_state_dict({k: t(v.copy()) for k, v in init.items()})
        for a, b in rings:
            run.storage.add_transitions_dagger(a, b)
        torch.manual_seed(4242)
        run.log_dict = {}
        run.update(1)
        cur = torch.cat([_hash(p) for p in run.student.parameters()] + [torch.tensor([float(run.log_dict["Train/dagger_loss"])], device=DEV).view(torch.int32).to(torch.int64)]).cpu()
        if first is None:
            first = cur
        elif not torch.equal(cur, first):
            bad += 1
            print(f"dagger rep {r} (prefetch {os.environ['PARTMANIP_GEOM_PREFETCH']}): parameters / loss differ from rep 0")
        del run
    print(f"dagger.update: {args.reps} repetitions (prefetch alternating on / off) in {time.time() - t0:.1f} s, {bad} differed")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--mode", default="both", choices=["fused", "materialised", "both"])
    ap.add_argument("--noise", action="store_true")
    ap.add_argument("--perlaunch", action="store_true")
    ap.add_argument("--dagger", action="store_true")
    args = ap.parse_args()
    print(f"stress_sparse_unet: {vars(args)} AMD_SERIALIZE_KERNEL={os.environ.get('AMD_SERIALIZE_KERNEL')}")
    bad = stress_dagger(args) if args.dagger else stress_backbone(args)
    print("repetitions that differed:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
