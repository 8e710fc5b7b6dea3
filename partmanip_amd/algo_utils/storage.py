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

    def _ring_write(self, stu_rows, tea_rows):
        """Append k rows to the DAgger ring at `mix_buf_ind`, wrapping modulo its size (one or two block copies)."""
        size, k = self.n_steps * self.num_envs, stu_rows.shape[0]
        at = self.mix_buf_ind
        head = min(k, size - at)
        for ring, rows in ((self.observations, stu_rows), (self.tea_obs, tea_rows)):
            ring[at:at + head].copy_(rows[:head])
            if head < k:                                  # wrapped: the remainder restarts at row 0
                ring[:k - head].copy_(rows[head:])
        self.mix_buf_ind = (at + k) % size
        self.cur_buf_size = min(size, self.cur_buf_size + k)

    def add_transitions_offline(self, folder, device, add_proprio_obs=False):
        """storage.py:58-82: every `scene_*/step_*.npy` shard {tsdf, proprio_state, tea_obs} becomes one ring row, in
        sorted (scene, step) order.  The shards are stacked on the host and land in the ring as block copies."""
        print('Read offline data from ', folder)
        scenes = sorted(os.listdir(folder))
        steps = sorted(os.listdir(pjoin(folder, scenes[0])))
        size = self.n_steps * self.num_envs
        stu, tea = [], []
        for scene in scenes:
            for step in steps:
                shard = np.load(pjoin(folder, scene, step), allow_pickle=True).item()
                row = np.asarray(shard['tsdf'], dtype=np.float32).reshape(-1)
                if add_proprio_obs:
                    row = np.concatenate([row, np.asarray(shard['proprio_state'], dtype=np.float32).reshape(-1)])
                stu.append(row)
                tea.append(np.asarray(shard['tea_obs'], dtype=np.float32).reshape(-1))
        stu, tea = torch.from_numpy(np.stack(stu)).to(device), torch.from_numpy(np.stack(tea)).to(device)
        for lo in range(0, stu.shape[0], size):           # more shards than ring rows: later ones overwrite, as row by row
            self._ring_write(stu[lo:lo + size], tea[lo:lo + size])
        self.last_episode_buf_ind = self.mix_buf_ind

    def add_transitions_dagger(self, stu_obs, tea_obs):
        """storage.py:84-91: the N rows of one env step go to the ring at `mix_buf_ind`."""
        self._ring_write(stu_obs, tea_obs)

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
