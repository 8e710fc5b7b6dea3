"""Synthetic / replayed rollout feeder: the vec-env duck-type the runners consume
(`reset() -> {mode: (N,D)}`, `step(a) -> (obs_dict, rew, reset, extras)`,
attrs num_envs / num_obs / num_actions / max_episode_length / reset_succ / rew_buf / success /
progress_buf / train_test_flag / dagger_reward_reset) standing in for the closed, CUDA-only
Isaac Gym stepper (tasks/hand_base.py:252-290), so that the hot path is the learner.

Observations are generated on the device from a seeded generator with the shapes and
distributions of SURVEY.md §8d: state obs ~ N(0,1); clouds = 1024 points ~ U([-1,1]^3) plus a
per-env translation U(-0.5,0.5); rewards ~ N(0,1); dones ~ Bernoulli(0.02); succ = done & B(0.5).
"""
import torch


class FeederEnv:
    def __init__(self, num_envs, num_obs, num_actions, device, seed=1234, max_episode_length=200, done_p=0.02,
                 point_num=1024):
        self.num_envs, self.num_obs, self.num_actions = num_envs, dict(num_obs), num_actions
        self.device = torch.device(device)
        self.max_episode_length = max_episode_length
        self.done_p, self.point_num = done_p, point_num
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.train_test_flag = 'train'
        self.dagger_reward_reset = None
        N = num_envs
        self.reset_succ = torch.zeros(N, dtype=torch.bool, device=self.device)
        self.success = torch.zeros(N, dtype=torch.bool, device=self.device)
        self.rew_buf = torch.zeros(N, device=self.device)
        self.progress_buf = torch.zeros(N, dtype=torch.long, device=self.device)

    def _obs(self):
        out = {}
        N = self.num_envs
        for mode, dim in self.num_obs.items():
            if dim == 0:
                out[mode] = torch.zeros(N, 0, device=self.device)
            elif mode == 'depth_sparse' and dim >= self.point_num * 4:
                out[mode] = self._sparse_voxels(dim)
            elif dim >= self.point_num * 3 and (dim % self.point_num) < 64 and mode != 'normal_state':
                c, tail = dim // self.point_num, dim % self.point_num
                pts = torch.rand(N, self.point_num, c, device=self.device, generator=self.gen) * 2 - 1
                pts = pts + (torch.rand(N, 1, c, device=self.device, generator=self.gen) - 0.5)
                parts = [pts.reshape(N, -1)]
                if tail:
                    parts.append(torch.randn(N, tail, device=self.device, generator=self.gen))
                out[mode] = torch.cat(parts, dim=1).contiguous()
            else:
                out[mode] = torch.randn(N, dim, device=self.device, generator=self.gen)
        return out

    def _sparse_voxels(self, dim, grid=50):
        """The 'depth_sparse' observation (tasks/hand_base.py:335-336; utils/depth2tsdf.py:88-120): `point_num` rows
        (x, y, z, tsdf) with integer voxel coordinates -- here the cells of a two-voxel-thick band around a random tilted
        plane through the grid (distinct cells, like a surface seen by the depth cameras), tsdf in (-0.2, 0.2)."""
        N, P, g = self.num_envs, self.point_num, self.gen
        layers = max(2, -(-P // (grid * grid)) + 1)
        score = torch.rand(N, grid * grid * layers, device=self.device, generator=g)
        pick = score.topk(P, dim=1).indices                              # P distinct (column, layer) pairs per env
        col, lay = pick % (grid * grid), pick // (grid * grid)
        xs, ys = col // grid, col % grid
        ab = torch.rand(N, 3, device=self.device, generator=g)
        z = (0.5 * ab[:, :1] * xs + 0.5 * ab[:, 1:2] * ys + ab[:, 2:3] * (grid / 4)).floor().long() + lay
        z = z.clamp_(0, grid - 1)
        f = torch.rand(N, P, device=self.device, generator=g) * 0.4 - 0.2
        rows = torch.stack([xs.float(), ys.float(), z.float(), f], dim=-1).reshape(N, P * 4)
        tail = dim - P * 4
        if tail:
            rows = torch.cat([rows, torch.randn(N, tail, device=self.device, generator=g)], dim=1)
        return rows.contiguous()

    def reset(self):
        self.progress_buf.zero_()
        return self._obs()

    def step(self, actions, save_image_path=None):
        N = self.num_envs
        assert actions.shape == (N, self.num_actions)
        rew = torch.randn(N, device=self.device, generator=self.gen)
        done = torch.rand(N, device=self.device, generator=self.gen) < self.done_p
        succ = done & (torch.rand(N, device=self.device, generator=self.gen) < 0.5)
        self.rew_buf, self.reset_succ, self.success = rew, succ, succ
        self.progress_buf = torch.where(done, torch.zeros_like(self.progress_buf), self.progress_buf + 1)
        extras = {'succ_rate': succ.float().mean().reshape(1)}
        return self._obs(), rew, done, extras

    def save_scene_pose(self, path):
        return {}


class ScreenLogger:
    """Screen-only stand-in for utils/logger.py (same four attributes / methods the runners use)."""

    def __init__(self, root='./logs', group='feeder', name='run', quiet=False):
        import os
        self.save_ckpt_dir = os.path.join(root, 'ckpts', group, name)
        self.save_pose_dir = os.path.join(root, 'scene_pose', group, name)
        self.save_video_dir = os.path.join(root, 'video', group, name)
        self.quiet = quiet

    def info(self, record_dict, iteration):
        if self.quiet:
            return
        print('#' * 80)
        print(f" Learning iteration {iteration} ".center(80))
        for k, v in record_dict.items():
            print(f"{k:<35}: {float(v):.6f}")

    def update_resume_path(self, resume_path):
        import os
        return os.path.join(os.path.dirname(self.save_ckpt_dir), resume_path)
