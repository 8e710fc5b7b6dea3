"""DAgger runner with the reference's interface (algorithms/dagger.py: `dagger(vec_env, cfg,
logger)`, `run()`, `update(it)`, `eval()`, `save(it)`, `resume(path)`, `load_pretrain(path)`).
The distillation update (dagger.py:299-337) runs on the HIP path: frozen-teacher forward,
student forward, fused tanh-MSE loss fwd+bwd (K11), student backward, fused Adam -- and, like
the PPO learner, without a host sync per mini-batch.
"""
import os
import time
from copy import deepcopy
from os.path import join as pjoin

import numpy as np
import torch

from ..algo_utils import RolloutStorage, ActorCritic, FusedAdam
from .. import ops, dist as pdist

try:
    from utils import path2video            # noqa: F401
except Exception:
    def path2video(*a, **k):
        return None


class dagger:
    def __init__(self, vec_env, cfg, logger):
        self.vec_env = vec_env
        self.num_envs = cfg['num_envs']
        self.stu_obs_mode = cfg['obs_mode']
        self.stu_num_obs = vec_env.num_obs[self.stu_obs_mode]
        self.stu_input_obs = self.stu_num_obs
        self.num_actions = vec_env.num_actions
        self.max_episode_length = vec_env.max_episode_length

        self.model_cfg = cfg['model']
        self.max_iter = cfg['max_iterations']
        self.n_steps = cfg['n_steps']
        self.n_updates = cfg['n_updates']
        self.num_mini_batches = cfg['n_minibatches']
        self.device = cfg['device']
        self.buf_size = cfg['buf_size']
        self.reward_reset = cfg['reward_reset']
        # per-step teacher reward curve for the reward-gap reset (dagger.py:33-34,234-237)
        if os.path.exists('teacher_reward.npy'):
            self.tea_rew = torch.tensor(np.load('teacher_reward.npy')).to(self.device)
        elif self.reward_reset:
            raise FileNotFoundError("teacher_reward.npy (dagger.py:33) is required when reward_reset is on")
        self.add_proprio_obs = cfg['add_proprio_obs']
        self.offline_data_pth = cfg['offline_data_pth']

        self.eval_round = cfg['eval_round']
        self.eval_freq = cfg['eval_frequence']
        self.save_freq = cfg['save_frequence']
        self.test_only = cfg['test_only']
        self.save_pose = cfg['save_pose']
        self.save_video = cfg['save_video']
        self.save_ckpt_dir = logger.save_ckpt_dir

        self.lr_schedule = cfg['lr_schedule']
        self.lr = cfg['lr']

        self.proprio_shape = cfg['add_proprio_obs'] * vec_env.num_obs['proprio_state']
        self.student = ActorCritic(self.stu_input_obs, self.num_actions, self.model_cfg, self.proprio_shape).to(self.device)
        f = self.student.flat()
        # Adam over student.parameters() (dagger.py:56); only the actor ever receives gradients
        self.optimizer = FusedAdam(f['actor'], f['grad_actor'][:f['n_actor'] + self.num_actions],
                                   [list(self.student.parameters())], lr=self.lr)

        self.logger = logger
        self.total_envsteps = 0
        self.total_time = 0
        self.curr_iter = 0

        self.teacher_path = cfg['teacher']
        assert self.teacher_path is not None and os.path.exists(self.teacher_path)
        print(f'load teacher ckpt from {self.teacher_path}!')
        tea = torch.load(self.teacher_path, map_location=self.device, weights_only=False)
        self.tea_obs_mode = tea['obs_mode']
        self.tea_num_obs = vec_env.num_obs[self.tea_obs_mode]
        self.teacher = ActorCritic(self.tea_num_obs, self.num_actions, tea['model_cfg']).to(self.device)
        self.teacher.load_state_dict(tea["model_state_dict"])
        assert tea['tricks']['use_state_norm'] == False  # noqa: E712 (dagger.py:73)

        self.resume(cfg['resume'])
        self.load_pretrain(cfg['pretrain'])

        self.storage = RolloutStorage(self.num_envs, self.buf_size, self.stu_num_obs, self.num_actions, self.device,
                                      sampler=cfg['sampler'], tea_obs_shape=self.tea_num_obs,
                                      max_length=self.max_episode_length)
        self.sync = pdist.maybe_sync(name="student")
        if self.sync is not None and self.sync.world > 1:       # the first collective, time-boxed and agreed over the store (dist.py)
            self.sync.mode = "eager"
            bad = self.sync.probe(f['actor'].device)
            if bad is not None:
                raise pdist.CollectiveError("data-parallel DAgger cannot start: " + bad)
        self._stage = {}
        self._loss_sum = torch.zeros(1, device=f['actor'].device)
        if self.sync is not None:                        # one student: rank 0's parameters / Adam state on every replica
            for t in (f['actor'], f['critic'], self.optimizer.m, self.optimizer.v, self.optimizer.state_dev):
                self.sync.broadcast_(t)

    # ------------------------------------------------------------------ checkpoints (dagger.py:81-120)
    def save(self, it):
        if self.sync is not None and self.sync.rank != 0:
            return                                       # replicas are identical: rank 0 writes the checkpoint
        os.makedirs(self.save_ckpt_dir, exist_ok=True)
        save_path = pjoin(self.save_ckpt_dir, f'model_{it}.pth')
        n_actor_params = len(list(self.student.actor.parameters()))
        torch.save({
            'iteration': it,
            'model_state_dict': {k: v.clone() for k, v in self.student.state_dict().items()},
            'optimizer_state_dict': self.optimizer.state_dict(active=set(range(1, 1 + n_actor_params))),
            'total_steps': self.total_envsteps,
            'obs_mode': self.stu_obs_mode,
            'teacher': self.teacher_path,
        }, save_path)
        print(f'save ckpt to {save_path}!')

    def load_pretrain(self, ckpt_path):
        if ckpt_path is not None:
            print(f'load pretrained ckpt from {ckpt_path}!')
            assert os.path.exists(ckpt_path)
            ckpt = torch.load(ckpt_path, map_location=self.device, weights_only=False)
            ckpt['model_state_dict'].pop('log_std')
            self.student.load_state_dict(ckpt["model_state_dict"], strict=False)

    def resume(self, ckpt_path):
        if ckpt_path is not None:
            print(f'load student ckpt from {ckpt_path}!')
            assert os.path.exists(ckpt_path)
            ckpt = torch.load(ckpt_path, map_location=self.device, weights_only=False)
            self.student.load_state_dict(ckpt["model_state_dict"])
            self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
            self.curr_iter = ckpt["iteration"]
            self.total_envsteps = ckpt["total_steps"]

    # ------------------------------------------------------------------ learner (dagger.py:299-337)
    def _rows(self, name, src, indices):
        n = len(indices)
        if self.storage.sampler == "sequential":
            return src[indices[0]:indices[0] + n]
        idx = torch.tensor(indices, dtype=torch.int64).to(src.device, non_blocking=True)
        buf = self._stage.get(name)
        if buf is None or buf.shape[0] != n:
            buf = self._stage[name] = torch.empty(n, src.shape[1], device=src.device)
        ops.gather_rows(src, idx, buf)
        return buf

    # ---- geometry of the NEXT mini-batch on a side stream (SparseUNet student), OPT-IN (PARTMANIP_GEOM_PREFETCH=1): the voxel
    # tables depend on the rows' coordinates only and are 2.9 of a 51.7 ms step at cfg 5, so they can be built for mini-batch
    # k + 1 while mini-batch k computes.  Bit-identical, and measured WITHOUT gain (round 3, tools/time_geometry_overlap.py:
    # forward + backward 48.1 ms with the tables ready, 51.7 inline, 51.1 with them on a side stream; bench.py cfg 5: 1228 vs
    # 1226 env-steps/s): the table kernels are not idle-latency work, their 8192 work-groups take CU slots and L2 bandwidth
    # from the gathered GEMMs for as long as they run.  Two staging buffers alternate; the side stream waits for the step that
    # last read the buffer it overwrites, the main stream for the tables; every table is handed to the main stream's
    # allocator bookkeeping (record_stream) before use.
    def _geom_prefetch(self, obs_all, indices, slot, after):
        side = self._side_stream
        main = torch.cuda.current_stream()
        if after is not None:
            side.wait_event(after)
        with torch.cuda.stream(side):
            idx = torch.tensor(indices, dtype=torch.int64).to(obs_all.device, non_blocking=True)
            key = ('stu_pf', slot)
            buf = self._stage.get(key)
            if buf is None or buf.shape[0] != len(indices):
                buf = self._stage[key] = torch.empty(len(indices), obs_all.shape[1], device=obs_all.device)
            ops.gather_rows(obs_all, idx, buf)
            g = self.student.actor.geometry(buf)
            ev = torch.cuda.Event()
            ev.record(side)

        def hand_over(o):
            if isinstance(o, torch.Tensor):
                o.record_stream(main)
            elif isinstance(o, dict):
                for v in o.values():
                    hand_over(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    hand_over(v)
        hand_over(g)
        idx.record_stream(side)
        return buf, g, ev

    def update(self, it):
        if self.storage.cur_buf_size < 16:
            return
        stu, tea = self.student, self.teacher
        f = stu.flat()
        tea.flat()
        n_a, A = f['n_actor'], self.num_actions
        if not self.optimizer.bound_to(f['actor']):
            self.optimizer.rebind(f['actor'], f['grad_actor'][:n_a + A])
        scal = f['scal_actor']
        stu_tanh, tea_tanh = stu.action_activate == 'tanh', tea.action_activate == 'tanh'
        # dagger.py:310 `teacher.act(tea_obs)` squashes with the TEACHER's action_activate / clipAction (its checkpoint's
        # model_cfg).  When both networks squash alike, K11 squashes the two raw means in one pass; otherwise the
        # teacher's action is formed first and K11 runs in action-target mode (bit 1, as bc.py does).
        same_squash = stu_tanh == tea_tanh and (not stu_tanh or stu.max_action == tea.max_action)
        mode = int(stu_tanh) | (0 if same_squash else 2)
        self._loss_sum.zero_()
        count = 0
        obs_all = self.storage.observations.view(-1, self.storage.observations.size(-1))
        tea_all = self.storage.tea_obs.view(-1, self.storage.tea_obs.size(-1))
        prefetch = (hasattr(stu.actor, 'take_geometry') and self.storage.sampler != "sequential" and obs_all.is_cuda
                    and os.environ.get("PARTMANIP_GEOM_PREFETCH", "0") == "1")
        if prefetch and getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream()
        step_no, done_ev = 0, [None, None]                 # done_ev[slot]: the step that last read staging buffer `slot` has been enqueued
        for _ in range(self.n_updates):
            batch = self.storage.mini_batch_generator(self.num_mini_batches)   # fresh sampler per epoch (dagger.py:305)
            it_b = iter(batch)
            nxt = next(it_b, None)
            pending = None
            while nxt is not None:
                indices, nxt = nxt, next(it_b, None)
                in_place = getattr(stu.actor, 'supports_row_index', False) and self.storage.sampler != "sequential"
                if pending is not None:                    # rows and tables of this mini-batch were built under the previous one
                    stu_obs, g_ready, ev = pending
                    torch.cuda.current_stream().wait_event(ev)
                    stu.actor.take_geometry(g_ready)
                    pending = None
                else:
                    stu_obs = None if in_place else self._rows('stu', obs_all, indices)
                tea_obs = self._rows('tea', tea_all, indices)
                with torch.no_grad():
                    tea_mu = tea.actor.hip_forward(tea_obs)                  # teacher.act, squashing fused into K11
                    if not same_squash:
                        tea_mu = tea_mu.contiguous()
                        ops.action_activation(tea_mu, tea_mu, tea.max_action, tea_tanh)
                if in_place:           # the backbone reads its rows of the ring where they lie (Conv3DNet: 0.8 GB per batch)
                    idx = torch.tensor(indices, dtype=torch.int64).to(obs_all.device, non_blocking=True)
                    stu_mu = stu.actor.hip_forward(obs_all, rows=idx)
                else:
                    stu_mu = stu.actor.hip_forward(stu_obs)
                dstu = torch.empty_like(stu_mu)
                ops.mse_tanh_loss(stu_mu, tea_mu, stu.max_action, mode, 1.0, scal, dstu)
                stu.actor.hip_backward(dstu)
                if self.sync:
                    self.sync.mean_(f['grad_actor'])
                self._loss_sum += scal[0:1]
                self.optimizer.step(n=n_a, n_clip=0, max_norm=0.0)           # no clipping (dagger.py:317-319)
                count += 1
                if prefetch:
                    slot = step_no & 1
                    done_ev[slot] = torch.cuda.Event()
                    done_ev[slot].record()
                    step_no += 1
                    if nxt is not None:                    # (this step is enqueued: the host may now block in the tables' two size reads)
                        pending = self._geom_prefetch(obs_all, nxt, step_no & 1, done_ev[step_no & 1])
        mean_loss = float(self._loss_sum.item()) / count
        print('update loss', mean_loss)
        if self.lr_schedule == 'linear_decay':
            lr_now = self.lr * max(1 - it / self.max_iter * 1.8, 0.1)
            for g in self.optimizer.param_groups:
                g['lr'] = lr_now
        elif self.lr_schedule != 'fixed':
            raise NotImplementedError
        self.log_dict['Train/learning_rate'] = self.optimizer.param_groups[0]['lr']
        self.log_dict['Train/dagger_loss'] = mean_loss

    # ------------------------------------------------------------------ rollout / eval (simulator-bound)
    def use_info_update_logdict(self, info_lst, mode):
        """dagger.py:280-297: Train = overwrite; Val/Test = running average over eval rounds."""
        for key in info_lst[0].keys():
            assert len(info_lst[0][key].shape) == 1, f"{key}: {info_lst[0][key].shape}"
            allv = torch.stack([info[key].float() for info in info_lst], dim=-1)
            mean, mx = torch.mean(allv), torch.mean(allv.max(dim=-1)[0])
            if mode != 'Train':
                self.log_dict[f'{mode}/{key}_mean'] = self.log_dict.get(f'{mode}/{key}_mean', 0) + mean / self.eval_round
                self.log_dict[f'{mode}/{key}_max'] = self.log_dict.get(f'{mode}/{key}_max', 0) + mx / self.eval_round
            else:
                self.log_dict[f'{mode}/{key}_mean'] = mean
                self.log_dict[f'{mode}/{key}_max'] = mx

    def eval(self):
        """dagger.py:122-178."""
        self.student.eval()
        if self.test_only:
            self.log_dict = {}
        with torch.no_grad():
            for r in range(self.eval_round):
                ep_infos, poses = [], []
                stu_obs = self.vec_env.reset()[self.stu_obs_mode]
                for i in range(self.max_episode_length):
                    actions = self.student.act(stu_obs)
                    img = pjoin(self.logger.save_video_dir, f"Iter{self.curr_iter}", f"{i}.png") if self.save_video else None
                    next_obs, rews, dones, infos = self.vec_env.step(actions, save_image_path=img)
                    infos['action_t'] = actions[:, :3].mean(dim=-1)
                    infos['action_r'] = actions[:, 3:6].mean(dim=-1)
                    infos['action_gripper'] = actions[:, -1]
                    infos['reward'] = rews
                    ep_infos.append(deepcopy(infos))
                    if self.save_pose:
                        d = self.vec_env.save_scene_pose(pjoin(self.logger.save_pose_dir, f"Iter{self.curr_iter}", f"{i}.npy"))
                        d['state'], d['action'] = stu_obs.cpu().numpy(), actions.cpu().numpy()
                        poses.append(deepcopy(d))
                    stu_obs = next_obs[self.stu_obs_mode]
                if self.save_pose:
                    for i, d in enumerate(poses):
                        d['success'] = ep_infos[-1]['obj_up_flag'].cpu().numpy()
                        np.save(pjoin(self.logger.save_pose_dir, f"Iter{self.curr_iter}", f"{i}.npy"), d)
                if self.save_video and r == self.eval_round - 1:
                    path2video(pjoin(self.logger.save_video_dir, f"Iter{self.curr_iter}"))
                self.use_info_update_logdict(ep_infos, 'Test' if self.test_only else 'Val')

    def run(self):
        """dagger.py:180-278."""
        if self.test_only:
            self.eval()
            self.logger.info(self.log_dict, self.curr_iter)
            return
        if self.offline_data_pth is not None:
            self.storage.add_transitions_offline(self.offline_data_pth, self.device, self.add_proprio_obs)
        obs = self.vec_env.reset()
        tea_obs, stu_obs = obs[self.tea_obs_mode], obs[self.stu_obs_mode]
        while self.curr_iter < self.max_iter:
            self.curr_iter += 1
            self.student.train()
            self.teacher.eval()
            self.log_dict = {}
            ep_infos = []
            t0 = time.time()
            for _ in range(self.n_steps):
                actions = self.student.random_act(stu_obs)
                next_obs, rews, dones, infos = self.vec_env.step(actions)
                self.storage.add_transitions_dagger(stu_obs, tea_obs)
                infos['action_t'] = actions[:, :3].mean(dim=-1)
                infos['action_r'] = actions[:, 3:6].mean(dim=-1)
                infos['action_gripper'] = actions[:, -1]
                tea_obs, stu_obs = next_obs[self.tea_obs_mode], next_obs[self.stu_obs_mode]
                ep_infos.append(deepcopy(infos))
                if self.reward_reset:            # reset envs whose reward lags the teacher's curve
                    lag = 10
                    prog = self.vec_env.progress_buf
                    self.vec_env.dagger_reward_reset = (prog > lag) & (rews < self.tea_rew[prog - lag])
            torch.cuda.synchronize()
            collection_time = time.time() - t0

            t0 = time.time()
            self.update(self.curr_iter)
            torch.cuda.synchronize()
            learn_time = time.time() - t0

            self.total_envsteps += self.n_steps * self.vec_env.num_envs
            self.total_time += collection_time + learn_time
            self.log_dict['Progress/total_steps'] = self.curr_iter
            self.log_dict['Progress/collection_time'] = collection_time
            self.log_dict['Progress/learn_time'] = learn_time
            self.log_dict['Progress/FPS'] = int(self.n_steps * self.vec_env.num_envs / (collection_time + learn_time))
            self.log_dict['Train/mean_action_noise_std'] = self.student.log_std.exp().mean().item()
            self.log_dict['Train/cur_buf_size'] = self.storage.cur_buf_size
            self.log_dict['Train/succ_buf_ind'] = self.storage.succ_buf_ind
            self.log_dict['Train/mix_buf_ind'] = self.storage.mix_buf_ind
            self.use_info_update_logdict(ep_infos, 'Train')

            if self.curr_iter % self.eval_freq == 0:
                self.eval()
                obs = self.vec_env.reset()
                tea_obs, stu_obs = obs[self.tea_obs_mode], obs[self.stu_obs_mode]
            if self.curr_iter % self.save_freq == 0:
                self.save(self.curr_iter)
            self.logger.info(self.log_dict, self.curr_iter)
