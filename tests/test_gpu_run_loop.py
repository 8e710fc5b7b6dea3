"""The runners' full `run()` loops (rollout -> learn -> log -> save) against the feeder env,
and the train.py command line, on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _ppo_cfg(net, obs_mode, N, T, state_norm):
    return dict(num_envs=N, obs_mode=obs_mode, succ_value=None,
                model=dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net),
                max_iterations=2, n_steps=T, n_updates=2, n_minibatches=4, device=DEV, eval_round=1, eval_frequence=2,
                save_frequence=2, test_only=False, save_pose=False, save_video=False, lr_schedule="linear_decay",
                lr=1e-4, desired_kl=0.1, epsilon_clip=0.2, gamma=0.99, lam=0.95,
                tricks=dict(mini_adv_norm=True, whole_adv_norm=True, use_state_norm=state_norm,
                            use_clipped_value_loss=True, use_grad_clip=True, max_grad_norm=0.5),
                sampler="random", resume=None)


@pytest.mark.parametrize("kind", ["mlp", "pointnet"])
def test_ppo_run_loop(kind, tmp_path):
    from partmanip_amd.algorithms import ppo
    from partmanip_amd.feeder import FeederEnv, ScreenLogger
    if kind == "mlp":
        net, mode, dim = dict(name="MLP", hid_dim=[64, 64], activation="tanh"), "normal_state", 53
    else:
        net, mode, dim = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=True), "depth_pc", 3072
    env = FeederEnv(16, {mode: dim}, 10, DEV, seed=5, max_episode_length=6)
    logger = ScreenLogger(str(tmp_path), "g", "n", quiet=True)
    run = ppo(env, _ppo_cfg(net, mode, 16, 4, kind == "mlp"), logger)
    before = torch.cat([p.detach().reshape(-1).clone() for p in run.actor_critic.parameters()])
    run.run()
    after = torch.cat([p.detach().reshape(-1) for p in run.actor_critic.parameters()])
    assert run.curr_iter == 2 and run.total_envsteps == 2 * 4 * 16
    assert torch.isfinite(after).all() and not torch.equal(before, after)
    for k in ("Train/surrogate_loss", "Train/value_function_loss", "Train/kl", "Progress/learn_time", "Val/succ_rate_mean"):
        assert k in run.log_dict and np.isfinite(float(run.log_dict[k])), k
    assert os.path.exists(os.path.join(logger.save_ckpt_dir, "model_2.pth"))


def test_dagger_run_loop(tmp_path, monkeypatch):
    from partmanip_amd.algorithms import ppo, dagger
    from partmanip_amd.feeder import FeederEnv, ScreenLogger
    monkeypatch.chdir(tmp_path)
    np.save("teacher_reward.npy", np.linspace(0, 1, 300).astype(np.float32))
    obs = {"normal_state": 53, "depth_pc": 3072 + 7, "proprio_state": 7}
    env = FeederEnv(16, obs, 10, DEV, seed=7, max_episode_length=5)
    tlog = ScreenLogger(str(tmp_path), "t", "n", quiet=True)
    tea = ppo(env, _ppo_cfg(dict(name="MLP", hid_dim=[64, 64], activation="tanh"), "normal_state", 16, 2, False), tlog)
    tea.save(1)
    cfg = dict(num_envs=16, obs_mode="depth_pc",
               model=dict(action_std=0.1, action_activate="tanh", clipAction=1.0,
                          network=dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False)),
               max_iterations=3, n_steps=1, n_updates=2, n_minibatches=2, device=DEV, buf_size=4, reward_reset=True,
               add_proprio_obs=True, offline_data_pth=None, eval_round=1, eval_frequence=3, save_frequence=3,
               test_only=False, save_pose=False, save_video=False, lr_schedule="linear_decay", lr=1e-4,
               teacher=os.path.join(tlog.save_ckpt_dir, "model_1.pth"), resume=None, pretrain=None, sampler="random")
    logger = ScreenLogger(str(tmp_path), "d", "n", quiet=True)
    run = dagger(env, cfg, logger)
    run.run()
    assert run.curr_iter == 3 and run.storage.cur_buf_size == 48
    assert np.isfinite(float(run.log_dict["Train/dagger_loss"]))
    assert os.path.exists(os.path.join(logger.save_ckpt_dir, "model_3.pth"))
    # resume from the saved student checkpoint
    cfg2 = dict(cfg, resume=os.path.join(logger.save_ckpt_dir, "model_3.pth"))
    run2 = dagger(env, cfg2, logger)
    assert run2.curr_iter == 3
    for a, b in zip(run.student.state_dict().values(), run2.student.state_dict().values()):
        assert torch.equal(a, b)


def test_train_py_command_line(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--algocfg", "ppo_pointnet", "--taskcfg",
                          "open_drawer", "--exp_name", "cli", "--algo.num_envs", "32", "--algo.n_steps", "4",
                          "--algo.max_iterations", "2", "--algo.n_minibatches", "2", "--log.log_root", str(tmp_path)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Train/surrogate_loss" in out.stdout and "Learning iteration 2" in out.stdout


def _cli(args, cwd=ROOT):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train.py")] + args, capture_output=True, text=True,
                         timeout=900, cwd=cwd)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


def test_train_py_dagger_tsdf_and_bc_configs(tmp_path):
    """The reference's shipped DAgger default (`dagger_tsdf.yaml`: Conv3DNet student on 50^3 TSDF volumes, state
    teacher) and `bc.yaml` through the command line: a teacher is trained and saved by `--algocfg ppo`, then
    distilled; then behaviour cloning from shards on disk."""
    root = str(tmp_path)
    _cli(["--algocfg", "ppo", "--taskcfg", "open_drawer", "--exp_name", "tea", "--algo.num_envs", "16", "--algo.n_steps", "4",
          "--algo.max_iterations", "1", "--algo.n_minibatches", "2", "--algo.save_frequence", "1",
          "--algo.tricks.use_state_norm", "--log.log_root", root])   # bool flags toggle the yaml value: state norm OFF, as dagger.py:73 requires of its teacher
    ck = [os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if f == "model_1.pth"]
    assert len(ck) == 1
    out = _cli(["--algocfg", "dagger_tsdf", "--taskcfg", "open_drawer", "--exp_name", "stu", "--algo.num_envs", "16",
                "--algo.buf_size", "4", "--algo.max_iterations", "2", "--algo.n_minibatches", "2", "--algo.teacher", ck[0],
                "--log.log_root", root])
    assert "Train/dagger_loss" in out
    data = os.path.join(root, "bc_data", "scene_00000")
    os.makedirs(data)
    rng = np.random.default_rng(0)
    for i in range(12):
        np.save(os.path.join(data, f"step_{str(i).zfill(5)}.npy"),
                dict(tsdf=rng.uniform(-1, 1, 50 ** 3).astype(np.float32), action=rng.uniform(-0.9, 0.9, 10).astype(np.float32),
                     proprio_state=np.zeros(0, dtype=np.float32)), allow_pickle=True)
    out = _cli(["--algocfg", "bc", "--taskcfg", "open_drawer", "--exp_name", "bc", "--algo.max_iterations", "2",
                "--algo.n_minibatches", "3", "--algo.data_path", os.path.dirname(data), "--log.log_root", root])
    assert "Train/bc_loss" in out
