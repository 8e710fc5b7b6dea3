"""Behaviour-cloning runner with the reference's interface (algorithms/bc.py: `bc(vec_env, cfg, logger)`,
`run()`, `save(it)`, `resume(path)`): a student regresses recorded actions from offline shards
`data_path/scene_*/step_*.npy` = {tsdf, action, proprio_state} (bc.py:12-31).

MI355X-first: the reference re-reads every shard from disk through 10 DataLoader workers each epoch; here the
whole dataset is read ONCE and stays resident in HBM (288 GB), an epoch is a row gather per mini-batch (K3),
and the step is the DAgger path's kernels -- student forward, fused MSE loss fwd+bwd (K11, action-target mode),
student backward, fused Adam -- with one host sync per epoch (the mean loss).  The shuffled index batches are
drawn by the same `torch.utils.data.DataLoader(..., shuffle=True)` construction as bc.py:113-115 (over the row
indices, no workers), so the sample order follows the reference's RNG consumption.
"""
import os
from os.path import join as pjoin

import numpy as np
import torch

from ..algo_utils import ActorCritic, FusedAdam
from .. import ops


class Tsdf_Dataset(torch.utils.data.Dataset):
    """bc.py:12-31, minus the per-item disk read: rows are addressed by index, data lives on the device."""

    def __init__(self, data_path):
        super().__init__()
        self.data_path = data_path
        self.env_lst = os.listdir(data_path)
        self.env_num = len(self.env_lst)
        self.step_num = len(os.listdir(pjoin(data_path, 'scene_00000')))

    def path(self, index):
        return pjoin(self.data_path, f'{self.env_lst[index // self.step_num]}/step_{str(index % self.step_num).zfill(5)}.npy')

    def __getitem__(self, index):
        return index

    def __len__(self):
        return self.env_num * self.step_num

    def load_resident(self, device):
        """Read every shard once -> (tsdf (n, D), action (n, A), proprio_state (n, S)) device tensors."""
        rows = [np.load(self.path(i), allow_pickle=True).item() for i in range(len(self))]
        cat = lambda k: torch.from_numpy(np.stack([np.asarray(r[k], dtype=np.float32).reshape(-1) for r in rows])).to(device)
        return cat('tsdf'), cat('action'), cat('proprio_state')


class bc:
    # cfg key -> attribute name, exactly the public attributes the reference's runner exposes (bc.py:36-62)
    _CFG_ATTRS = dict(num_envs='num_envs', obs_mode='stu_obs_mode', model='model_cfg', max_iterations='max_iter',
                      device='device', data_path='data_path', n_minibatches='n_minibatches',
                      add_proprio_obs='add_proprio_obs', eval_round='eval_round', eval_frequence='eval_freq',
                      save_frequence='save_freq', test_only='test_only', save_pose='save_pose', save_video='save_video',
                      lr_schedule='lr_schedule', lr='lr')

    def __init__(self, vec_env, cfg, logger):
        for key, attr in self._CFG_ATTRS.items():
            setattr(self, attr, cfg[key])
        self.vec_env, self.logger = vec_env, logger
        self.stu_num_obs = vec_env.num_obs[self.stu_obs_mode]
        self.num_actions, self.max_episode_length = vec_env.num_actions, vec_env.max_episode_length
        self.save_ckpt_dir = logger.save_ckpt_dir
        proprio = self.add_proprio_obs * vec_env.num_obs['proprio_state']
        self.student = ActorCritic(self.stu_num_obs, self.num_actions, self.model_cfg, proprio).to(self.device)
        f = self.student.flat()
        # Adam over student.parameters() (bc.py:68); only the actor ever receives gradients
        self.optimizer = FusedAdam(f['actor'], f['grad_actor'][:f['n_actor'] + self.num_actions],
                                   [list(self.student.parameters())], lr=self.lr)
        self.total_time = self.curr_iter = 0
        self._loss_sum = torch.zeros(1, device=f['actor'].device)
        self._stage = {}
        self.resume(cfg['resume'])

    # ------------------------------------------------------------------ checkpoints (bc.py:79-107)
    def save(self, it):
        os.makedirs(self.save_ckpt_dir, exist_ok=True)
        save_path = pjoin(self.save_ckpt_dir, f'model_{it}.pth')
        n_actor_params = len(list(self.student.actor.parameters()))
        torch.save({
            'iteration': it,
            'model_state_dict': {k: v.clone() for k, v in self.student.state_dict().items()},
            'optimizer_state_dict': self.optimizer.state_dict(active=set(range(1, 1 + n_actor_params))),
            'obs_mode': self.stu_obs_mode,
            'total_steps': 0,
            'tricks': {'use_state_norm': False},
            'teacher': 0,
        }, save_path)
        print(f'save ckpt to {save_path}!')

    def resume(self, ckpt_path):
        if ckpt_path is not None:
            print(f'load student ckpt from {ckpt_path}!')
            assert os.path.exists(ckpt_path)
            ckpt = torch.load(ckpt_path, map_location=self.device, weights_only=False)
            self.student.load_state_dict(ckpt["model_state_dict"])
            self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
            self.curr_iter = ckpt["iteration"]
            assert ckpt['obs_mode'] == self.stu_obs_mode

    # ------------------------------------------------------------------ learner (bc.py:109-177)
    def _gather(self, name, src, idx):
        buf = self._stage.get(name)
        if buf is None or buf.shape[0] != idx.shape[0]:
            buf = self._stage[name] = torch.empty(idx.shape[0], src.shape[1], device=src.device)
        ops.gather_rows(src, idx, buf)
        return buf

    def train_epoch(self, data, loader):
        """One pass of bc.py:122-146 over the resident dataset; returns the mean mini-batch loss."""
        tsdf, actions, states = data
        stu = self.student
        f = stu.flat()
        n_a, scal = f['n_actor'], f['scal_actor']
        mode = 3 if stu.action_activate == 'tanh' else 0           # student squashed, target = recorded action
        self._loss_sum.zero_()
        count = 0
        for index_batch in loader:
            idx = index_batch.to(tsdf.device, non_blocking=True)
            x = self._gather('tsdf', tsdf, idx)
            if self.add_proprio_obs and states.shape[1] > 0:       # a task may declare a 0-wide proprio state
                x = torch.cat([x, self._gather('state', states, idx)], dim=-1)
            act = self._gather('action', actions, idx)
            stu_mu = stu.actor.hip_forward(x)
            dstu = torch.empty_like(stu_mu)
            ops.mse_tanh_loss(stu_mu, act, stu.max_action, mode, 1.0, scal, dstu)
            stu.actor.hip_backward(dstu)
            self._loss_sum += scal[0:1]
            self.optimizer.step(n=n_a, n_clip=0, max_norm=0.0)
            count += 1
        return float(self._loss_sum.item()) / count

    def run(self):
        if self.test_only:
            raise NotImplementedError
        dataset = Tsdf_Dataset(self.data_path)
        batch_size = len(dataset) // self.n_minibatches
        data = dataset.load_resident(self.device)
        loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=True, num_workers=0)
        while self.curr_iter < self.max_iter:
            self.curr_iter += 1
            self.log_dict = {}
            mean_loss = self.train_epoch(data, loader)
            if self.lr_schedule == 'linear_decay':
                lr_now = self.lr * (1 - self.curr_iter / self.max_iter)
            elif self.lr_schedule == 'step_decay':
                lr_now = self.lr if self.curr_iter < self.max_iter / 2 else self.lr * 0.1
            elif self.lr_schedule != 'fixed':
                raise NotImplementedError
            else:
                lr_now = None
            if lr_now is not None:
                for g in self.optimizer.param_groups:
                    g['lr'] = lr_now
            self.log_dict['Train/learning_rate'] = self.optimizer.param_groups[0]['lr']
            self.log_dict['Train/bc_loss'] = mean_loss
            self.log_dict['Progress/total_steps'] = self.curr_iter
            if self.curr_iter % self.save_freq == 0:
                self.save(self.curr_iter)
            self.logger.info(self.log_dict, self.curr_iter)
