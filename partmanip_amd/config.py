"""Config loading with the reference's semantics (utils/config.py:35-140) minus Isaac Gym:
three yaml files (cfg/base_cfg.yaml, cfg/tasks/<task>.yaml, cfg/algos/<algo>.yaml) merged into
one dict; every leaf becomes a typed `--a.b.c` command-line override (bools are toggles, `None`
leaves are strings, lists take `nargs=+`); derived keys are copied into cfg['task'] / cfg['algo']
exactly as utils/config.py:114-138 does (device, resume, test_only, model.clipAction,
succ_value, num_envs, learn_input_mode, add_proprio_obs, algo_name, task_name)."""
import os
from argparse import ArgumentParser
from collections import abc
from os.path import join as pjoin

import yaml


def add_args(parser, cfg, prefix=""):
    for k, v in cfg.items():
        flag = "--" + prefix + k
        if isinstance(v, bool):
            parser.add_argument(flag, default=None, action="store_false" if v else "store_true")
        elif isinstance(v, int):
            parser.add_argument(flag, type=int)
        elif isinstance(v, float):
            parser.add_argument(flag, type=float)
        elif isinstance(v, str) or v is None:
            parser.add_argument(flag)
        elif isinstance(v, dict):
            add_args(parser, v, prefix + k + ".")
        elif isinstance(v, abc.Iterable):
            parser.add_argument(flag, type=type(v[0]), nargs="+")
        else:
            print(f"WARNING: cannot parse key {prefix + k} of type {type(v)}")
    return parser


def process_cfgs(argv=None, root=None):
    root = root or os.getcwd()
    pre = ArgumentParser(add_help=False)
    pre.add_argument('--taskcfg', default='open_drawer')
    pre.add_argument('--algocfg', default='ppo')
    pargs, others = pre.parse_known_args(argv)

    def load(rel):
        with open(pjoin(root, rel), 'r') as f:
            return yaml.load(f, Loader=yaml.SafeLoader)
    cfg = load('cfg/base_cfg.yaml')
    cfg['task'] = load(f'cfg/tasks/{pargs.taskcfg}.yaml')
    cfg['algo'] = load(f'cfg/algos/{pargs.algocfg}.yaml')

    args = vars(add_args(ArgumentParser(description="partmanip-mi learner"), cfg).parse_args(others))
    for k, v in args.items():
        if v is None:
            continue
        node, path = cfg, k.split('.')
        for kk in path[:-1]:
            node = node[kk]
        print(f'overwrite {k} from {node[path[-1]]} to {v}!')
        node[path[-1]] = v

    cfg['device'] = 'cpu' if cfg['device_type'] == 'cpu' else f"{cfg['device_type']}:{cfg['device_id']}"
    for k in ('device_id', 'device', 'save_video'):
        cfg['task'][k] = cfg[k]
    for k in ('resume', 'test_only', 'device', 'save_pose', 'save_video', 'pretrain'):
        cfg['algo'][k] = cfg[k]
    cfg['algo']['model']['clipAction'] = cfg['task']['clipActions']
    cfg['algo']['succ_value'] = cfg['task']['succ_value']
    cfg['task']['num_envs'] = cfg['algo']['num_envs']
    cfg['task']['learn_input_mode'] = cfg['algo']['obs_mode']
    cfg['task']['add_proprio_obs'] = cfg['algo']['add_proprio_obs']
    cfg['algo_name'] = cfg['algo']['algo']
    cfg['task_name'] = cfg['task']['task']
    return cfg


def num_actions(task_cfg):
    """tasks/load_robot.py:15-30: 7 for ik control, +3 with the mobile base."""
    return 7 + (3 if task_cfg.get('robot', {}).get('mobile', False) else 0)
