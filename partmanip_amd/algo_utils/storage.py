"""RolloutStorage with the reference's constructor, attributes and methods
(algorithms/algo_utils/storage.py:7-138); `compute_returns` is the HIP GAE scan.

Layout in HBM: every PPO tensor is (T, N, D) contiguous with N (the env batch) next to the
feature dim, so a time-reversed per-env scan reads coalesced rows and a sequential
mini-batch is a contiguous slice of the flat (T*N, D) view (row = t*N + n).
"""
import os
from os.path import join as pjoin

import numpy as np
import torch
from torch.utils.data.sampler import BatchSampler, SequentialSampler, SubsetRandomSampler

from .. import ops


class RolloutStorage:

    def __init__(self, num_envs, n_steps, obs_shape, actions_shape, device, default_succ_value=0,
                 whole_adv_norm=False, sampler='sequential', tea_obs_shape=None, max_length=None):
        self.device = device
        self.sampler = sampler
        self.n_steps = n_steps
        self.num_envs = num_envs
        self.step = 0
        self.whole_adv_norm = whole_adv_norm
        self.max_episode_length = max_length
        self.first_fill = True
        self.default_succ_value = default_succ_value
        self._ws = None
        self._mom = None

        rows = self.n_steps * self.num_envs
        if tea_obs_shape is not None:
            # DAgger: one flat ring of `rows` (student obs, teacher obs) pairs -- storage.py:20-27
            self._alloc(dict(tea_obs=(rows, tea_obs_shape), observations=(rows, obs_shape), succ_flag=(rows, 1)))
            self.mix_buf_ind = 0
            self.succ_buf_ind = self.max_episode_length * self.num_envs
            self.cur_buf_size = 0
            self.last_episode_buf_ind = 0
        else:
            # PPO: (T, N, D) rollout tensors, names as the runner indexes them -- storage.py:28-41
            tn = (self.n_steps, num_envs)
            self._alloc(dict(observations=tn + (obs_shape,), rewards=tn + (1,), actions=tn + (actions_shape,),
                             actions_log_prob=tn + (1,), values=tn + (1,), returns=tn + (1,), advantages=tn + (1,),
                             mu=tn + (actions_shape,), sigma=tn + (actions_shape,), step_id=tn + (1,)))
            self._alloc(dict(dones=tn + (1,), succs=tn + (1,)), dtype=torch.bool)
            self.cur_buf_size = rows

    def _alloc(self, spec, dtype=torch.float32):
        for name, shape in spec.items():
            setattr(self, name, torch.zeros(*shape, dtype=dtype, device=self.device))

    def add_transitions(self, observations, actions, rewards, dones, succs, values, actions_log_prob, mu, sigma):
        """storage.py:43-56: write slot `step` of every rollout tensor; overflow raises like the reference."""
        t = self.step
        if t >= self.n_steps:
            raise AssertionError("Rollout buffer overflow")
        col = lambda v: v.view(-1, 1)
        for buf, val in ((self.observations, observations), (self.actions, actions), (self.rewards, col(rewards)),
                         (self.dones, col(dones)), (self.succs, col(succs)), (self.values, values),
                         (self.actions_log_prob, col(actions_log_prob)), (self.mu, mu), (self.sigma, sigma)):
            buf[t].copy_(val)
        self.step = t + 1

    def add_transitions_offline(self, folder, device, add_proprio_obs=False):
        """storage.py:58-82: scene_*/step_*.npy dicts {tsdf, proprio_state, tea_obs} row by row."""
        print('Read offline data from ', folder)
        scene_list = sorted(os.listdir(folder))
        step_list = sorted(os.listdir(pjoin(folder, scene_list[0])))
        max_buf_size = self.n_steps * self.num_envs
        for scene in scene_list:
            for step in step_list:
                data = np.load(pjoin(folder, scene, step), allow_pickle=True).item()
                tsdf = torch.tensor(data['tsdf']).reshape(-1).to(device)
                if add_proprio_obs:
                    stu_obs = torch.cat((tsdf, torch.tensor(data['proprio_state']).to(device)), dim=-1)
                else:
                    stu_obs = tsdf
                tea_obs = torch.tensor(data['tea_obs']).to(device)
                self.observations[self.mix_buf_ind:self.mix_buf_ind + 1].copy_(stu_obs)
                self.tea_obs[self.mix_buf_ind:self.mix_buf_ind + 1].copy_(tea_obs)
                self.mix_buf_ind = (self.mix_buf_ind + 1) % max_buf_size
                self.last_episode_buf_ind = self.mix_buf_ind
                if self.cur_buf_size < max_buf_size:
                    self.cur_buf_size += 1

    def add_transitions_dagger(self, stu_obs, tea_obs):
        """storage.py:84-91: N rows at mix_buf_ind, wrap modulo the ring size."""
        self.observations[self.mix_buf_ind:self.mix_buf_ind + self.num_envs].copy_(stu_obs)
        self.tea_obs[self.mix_buf_ind:self.mix_buf_ind + self.num_envs].copy_(tea_obs)
        max_buf_size = self.n_steps * self.num_envs
        self.mix_buf_ind = (self.mix_buf_ind + self.num_envs) % max_buf_size
        if self.cur_buf_size < max_buf_size:
            self.cur_buf_size += self.num_envs

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam, moments_sync=None):
        """storage.py:96-114 as one HIP scan (K1) + optional whole-batch normalisation (K2).
        `moments_sync(mom2) -> count` lets the data-parallel runner all-reduce {sum, sumsq}."""
        if self._ws is None:
            self._ws = ops.Workspace(self.rewards.device)
            self._mom = torch.zeros(2, dtype=torch.float64, device=self.rewards.device)
        ops.gae_scan(self.rewards, self.values, self.dones, self.succs, last_values.contiguous(), self.returns,
                     self.advantages, gamma, lam, self.default_succ_value)
        if self.whole_adv_norm:
            ops.moments(self.advantages, self._mom, self._ws)
            count = self.advantages.numel()
            if moments_sync is not None:
                count = moments_sync(self._mom, count)
            ops.normalize_apply(self.advantages, self._mom, count)

    def mini_batch_generator(self, num_mini_batches):
        """storage.py:125-138 verbatim semantics: re-iterable BatchSampler, 2048 cap, drop_last;
        'random' draws torch.randperm from the global (CPU) RNG on every iteration."""
        batch_size = self.cur_buf_size
        mini_batch_size = min(int(batch_size // num_mini_batches), 2048)
        if self.sampler == "sequential":
            subset = SequentialSampler(range(batch_size))
        elif self.sampler == "random":
            subset = SubsetRandomSampler(range(batch_size))
        return BatchSampler(subset, mini_batch_size, drop_last=True)
