#!/usr/bin/env python3
"""Entry point with the reference's command line (train.py:52-77):

    python train.py --algocfg ppo_pointnet --taskcfg open_drawer --exp_name run0 [--algo.lr 1e-4 ...]

The Isaac Gym tasks are replaced by the synthetic rollout feeder (partmanip_amd/feeder.py);
the runners are the MI355X learners behind the reference's `algorithms.ppo` / `dagger` API.
Multi-GPU: launch under `python -m torch.distributed.run --nproc-per-node N train.py ...`
(each rank takes num_envs / N envs; gradients are all-reduced over RCCL)."""
import os
import random

import numpy as np
import torch

from algorithms import ppo, dagger, bc  # noqa: F401  (resolved by name below, like the reference's eval())
from partmanip_amd import dist as pdist
from partmanip_amd.config import process_cfgs, num_actions
from partmanip_amd.feeder import FeederEnv, ScreenLogger


def pick_seed(seed, exp_name, resume):
    """The seed / run-name resolution of train.py:16-31 (no RNG is seeded yet)."""
    if 'seed' in exp_name:
        seed = int(exp_name.split('seed')[-1])
    elif resume is not None:
        try:
            seed = int(resume.split('/')[-2].split('seed')[-1])
        except Exception:
            seed = 1234
    elif seed == -1:
        seed = np.random.randint(0, 10000)
    if 'seed' not in exp_name:
        exp_name = exp_name + f'_seed{seed}'
    return int(seed), exp_name


def set_seed(seed, exp_name, resume):
    """train.py:16-50 minus the CUDA-specific determinism switches.  Under data parallelism rank 0 resolves the seed
    (`seed: -1` draws one) and every rank uses it: one model, one run name, one checkpoint directory."""
    seed, exp_name = pdist.resolve_seed(lambda: pick_seed(seed, exp_name, resume))
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    return seed, exp_name


def feeder_obs_dims(cfg):
    """Observation widths per mode, as tasks/hand_base.py:45-54 derives them: plain modes from the task yaml; the
    TSDF modes ('mesh_tsdf' / 'depth_tsdf') are resolution^3 volumes, plus the proprio state when the algo asks."""
    dims = {k: v for k, v in cfg['task']['obs_mode'].items() if not isinstance(v, dict)}
    mode = cfg['algo']['obs_mode']
    if mode not in dims:
        if 'tsdf' in mode and 'tsdf' in cfg['task']['obs_mode']:
            dims[mode] = cfg['task']['obs_mode']['tsdf']['resolution'] ** 3
        else:
            raise KeyError(f"obs_mode '{mode}' is not declared in the task yaml")
    if cfg['algo'].get('add_proprio_obs'):
        dims[mode] += dims.get('proprio_state', 0)
    return dims


def main():
    cfg = process_cfgs(root=os.path.dirname(os.path.abspath(__file__)))
    rank, world, local = pdist.init_from_env()
    if world > 1:
        cfg['device'] = cfg['algo']['device'] = f"cuda:{local}"
        lo, hi = pdist.shard_envs(cfg['algo']['num_envs'], rank, world)
        cfg['algo']['num_envs'] = cfg['task']['num_envs'] = hi - lo
    cfg['seed'], cfg['exp_name'] = set_seed(cfg['seed'], cfg['exp_name'], cfg['resume'])
    group = cfg['log']['group'] or f"{cfg['task_name']}_{cfg['algo_name']}"
    logger = ScreenLogger(cfg['log']['log_root'], group, cfg['log']['id'] or cfg['exp_name'], quiet=rank != 0)
    if cfg['resume'] is not None:
        cfg['algo']['resume'] = cfg['resume'] = logger.update_resume_path(cfg['resume'])
    if cfg['pretrain'] is not None:
        cfg['algo']['pretrain'] = cfg['pretrain'] = logger.update_resume_path(cfg['pretrain'])
    torch.cuda.set_device(cfg['device'])
    env = FeederEnv(cfg['algo']['num_envs'], feeder_obs_dims(cfg), num_actions(cfg['task']), cfg['device'],
                    seed=cfg['seed'] + rank, max_episode_length=cfg['task']['maxEpisodeLength'])
    runner = {'ppo': ppo, 'dagger': dagger, 'bc': bc}[cfg['algo_name']](env, cfg['algo'], logger)
    runner.run()
    return runner


if __name__ == '__main__':
    main()
