"""Shared fixture plumbing for the parity tests (oracle and HIP path alike)."""
import os

import numpy as np
import torch

from tests.golden import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------- observed parity margins
def margins_path():
    """Where the GPU tests append what they OBSERVED next to what they tolerate (one JSON object per line; gpurun merges
    gpurun_out/ back, tools/margins_summary.py folds the lines into profiles/parity_margins.json)."""
    return os.environ.get("PARTMANIP_MARGINS", os.path.join(ROOT, "gpurun_out", "parity_margins.jsonl"))


def record_margin(name, observed, tol, **extra):
    """observed / tol in the same unit (a relative error, an absolute error, a count ...); never raises."""
    import json
    try:
        path = margins_path()
        os.makedirs(os.path.dirname(path), exist_ok=True)
        test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
        with open(path, "a") as f:
            f.write(json.dumps(dict(test=test, name=name, observed=float(observed), tol=float(tol), **extra)) + "\n")
    except Exception:
        pass


def assert_close_rec(name, got, want, rtol, atol=0.0):
    """np.testing.assert_allclose that also records the worst element as a fraction of its tolerance."""
    g, w = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(g - w)
    bound = atol + rtol * np.abs(w)
    frac = float((err / np.maximum(bound, 1e-300)).max()) if err.size else 0.0
    rel = float((err / np.maximum(np.abs(w), 1e-300)).max()) if err.size else 0.0
    record_margin(name, frac, 1.0, unit="fraction of (atol + rtol*|want|)", rtol=rtol, atol=atol, worst_abs=float(err.max()) if err.size else 0.0,
                  worst_rel=rel)
    _ORIG_ALLCLOSE(g, w, rtol=rtol, atol=atol, err_msg=name)


_ORIG_ALLCLOSE = np.testing.assert_allclose                 # (tests/conftest.py wraps np.testing.assert_allclose for the GPU tests)


def recording_allclose(actual, desired, rtol=1e-7, atol=0, *args, **kw):
    """np.testing.assert_allclose + the observed margin of EVERY such assertion of a GPU test (named by its call site)."""
    import inspect
    try:
        fr = inspect.stack()[1]
        where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
        g, w = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
        g, w = np.broadcast_arrays(g, w)
        err = np.abs(g - w)
        bound = atol + rtol * np.abs(w)
        if err.size and np.all(np.isfinite(err)):
            frac = float((err / np.maximum(bound, 1e-300)).max())
            record_margin(f"assert_allclose @ {where}" + (f" [{kw['err_msg']}]" if kw.get("err_msg") else ""), frac, 1.0,
                          unit="fraction of (atol + rtol*|want|)", rtol=float(rtol), atol=float(atol), worst_abs=float(err.max()))
    except Exception:
        pass
    return _ORIG_ALLCLOSE(actual, desired, rtol, atol, *args, **kw)


def load_fixture(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def state_dict_t(sd_np):
    return {k: t(v.copy()) for k, v in sd_np.items()}


def ppo_model_cfg(c):
    return dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0, network=dict(c["net"]))


def ppo_cfg(c, device="cpu", num_envs=None):
    """The cfg dict the reference's `ppo.__init__` consumes (ppo.py:21-81)."""
    return dict(num_envs=num_envs or c["N"], obs_mode="normal_state", succ_value=c.get("succ_value"),
                model=ppo_model_cfg(c), max_iterations=c["max_iterations"], n_steps=c["T"],
                n_updates=c["n_updates"], n_minibatches=c["n_minibatches"], device=device, eval_round=1,
                eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False,
                save_video=False, lr_schedule=c["lr_schedule"], lr=c["lr"], desired_kl=c["desired_kl"],
                epsilon_clip=c["epsilon_clip"], gamma=c["gamma"], lam=c["lam"], tricks=dict(c["tricks"]),
                sampler=c["sampler"], resume=None)


def ppo_rollout(c, fx):
    """Rollout tensors of a PPO case: policy-independent ones rebuilt from `cases`,
    policy-dependent ones (values, old log-prob, old mu/sigma) from the fixture."""
    raw = cases.ppo_raw_inputs(c)
    st = dict(observations=t(raw["observations"]), actions=t(raw["actions"]), rewards=t(raw["rewards"]),
              dones=t(raw["dones"]), succs=t(raw["succs"]), values=t(fx["values"]),
              actions_log_prob=t(fx["actions_log_prob"]), mu=t(fx["mu"]), sigma=t(fx["sigma"]),
              last_values=t(fx["last_values"]))
    return st


def flat_state(sd):
    return np.concatenate([np.asarray(v.detach().cpu()).reshape(-1).astype(np.float32) for v in sd.values()])


def sd_order(c, proprio=0, net_key="net", o_key="O"):
    """Key order of the reference's ActorCritic.state_dict(): log_std, actor.*, critic.*"""
    sd = cases.actor_critic_state(c[net_key], c[o_key], c["A"], c["action_std"], c["seed"], proprio)
    return list(sd.keys())


class FakeEnv:
    """Smallest object satisfying the attribute reads of ppo.__init__/dagger.__init__."""

    def __init__(self, num_envs, num_obs, num_actions):
        self.num_envs, self.num_obs, self.num_actions = num_envs, num_obs, num_actions
        self.max_episode_length = 200


class FakeLogger:
    def __init__(self, d):
        self.save_ckpt_dir = d
        self.save_video_dir = d
        self.save_pose_dir = d

    def info(self, *a, **k):
        pass


def assert_params_close(got, want, lr, n_steps, q_tol=2e-5):
    """Parameter comparison that does not depend on the host CPU's sgemm summation order: Adam turns a gradient
    element that is ~0 at step 1 into a move of up to lr whatever its sign, so a handful of elements may differ by
    O(lr) between two correct fp32 evaluations: 99.9 % within q_tol, none beyond 2.5 * lr * steps."""
    diff = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64))
    assert np.quantile(diff, 0.999) < q_tol, np.quantile(diff, 0.999)
    assert diff.max() < 2.5 * lr * n_steps, diff.max()


def per_tensor_update_error(fin_flat, ref_flat, init_sd, stride=1):
    """For every tensor of the state dict: ||got - ref||_2 / ||ref - init||_2 over the sampled elements (those at flat
    positions = 0 mod `stride`, the way the fixtures subsample) -- the error RELATIVE TO HOW FAR THE UPDATE MOVED that
    tensor.  A tensor that moves the wrong way on every step scores >= 1 however small it is (b1, log_std, ...), which
    a whole-vector quantile or a loose max bound cannot see.  Unmoved tensors report the absolute error instead."""
    init = np.concatenate([np.asarray(v, dtype=np.float32).reshape(-1) for v in init_sd.values()])
    got = np.asarray(fin_flat, dtype=np.float64)[::stride]
    ref = np.asarray(ref_flat, dtype=np.float64)
    ini = init.astype(np.float64)[::stride]
    assert got.shape == ref.shape == ini.shape, (got.shape, ref.shape, ini.shape)
    out, off = {}, 0
    for k, v in init_sd.items():
        n = int(np.asarray(v).size)
        first = (-off) % stride                          # first sampled position inside [off, off + n)
        cnt = 0 if first >= n else (n - first + stride - 1) // stride
        lo = (off + first) // stride
        g, r, i0 = got[lo:lo + cnt], ref[lo:lo + cnt], ini[lo:lo + cnt]
        off += n
        if cnt == 0:
            continue
        moved, err = np.linalg.norm(r - i0), np.linalg.norm(g - r)
        out[k] = (err / moved, moved) if moved > 0 else (err, 0.0)
    return out


def assert_update_matches(fin_flat, ref_flat, init_sd, lr, n_steps, stride=1, rel=1.2e-2, q_lr=5e-3, max_lr_steps=0.35):
    """Whole-vector bounds (99.9 % quantile < q_lr * lr; max < max_lr_steps * lr * steps) PLUS the per-tensor relative-L2
    bound; what was observed is recorded beside each bound (profiles/parity_margins.json).  Defaults = 4x the worst value
    observed over the golden / autograd-bridge / SparseUNet cases on the MI355X box (round 3: 3.2e-3, 1.3e-3 lr, 0.085 lr
    steps; rounds 1-2 asserted 5e-2, 5e-2, 2.5)."""
    diff = np.abs(np.asarray(fin_flat, dtype=np.float64)[::stride] - np.asarray(ref_flat, dtype=np.float64))
    q999 = float(np.quantile(diff, 0.999))
    errs = per_tensor_update_error(fin_flat, ref_flat, init_sd, stride)
    worst = max((e for e, moved in errs.values() if moved > 0), default=0.0)
    worst_k = max(((e, k) for k, (e, moved) in errs.items() if moved > 0), default=(0.0, ""))[1]
    record_margin("update: 99.9% quantile of |param - ref| / lr", q999 / lr, q_lr, lr=lr, steps=n_steps)
    record_margin("update: max |param - ref| / (lr * steps)", float(diff.max()) / (lr * n_steps), max_lr_steps, lr=lr, steps=n_steps)
    record_margin("update: worst per-tensor ||got - ref|| / ||ref - init||", worst, rel, tensor=worst_k)
    assert q999 < q_lr * lr, (q999, lr)
    assert diff.max() < max_lr_steps * lr * n_steps, (diff.max(), lr)
    bad = {k: e for k, (e, moved) in errs.items() if (e > rel if moved > 0 else e > 1e-7)}
    assert not bad, f"per-tensor update error above {rel}: {bad}"
    return worst


def assert_flat_params_close(name, got, ref, lr, n_steps, q_lr=5e-2, max_lr_steps=2.5):
    """Two flat parameter vectors of the same run done two ways (ranks vs one process, ...): 99.9 % quantile and max bounds,
    observed values recorded."""
    diff = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
    q = float(np.quantile(diff, 0.999))
    record_margin(f"{name}: 99.9% quantile of |a - b| / lr", q / lr, q_lr)
    record_margin(f"{name}: max |a - b| / (lr * steps)", float(diff.max()) / (lr * n_steps), max_lr_steps)
    assert q < q_lr * lr and diff.max() < max_lr_steps * lr * n_steps, (q, diff.max(), lr)
