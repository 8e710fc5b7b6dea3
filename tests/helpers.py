"""Shared fixture plumbing for the parity tests (oracle and HIP path alike)."""
import os

import numpy as np
import torch

from tests.golden import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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
