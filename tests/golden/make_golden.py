#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/*.npz by RUNNING THE REFERENCE.

Dev-container only: imports `/root/reference` (read-only mount) with three
`sys.modules` stubs and drives the reference's own
`RolloutStorage.compute_returns`, `ppo.update`, `dagger.update`,
`ActorCritic.update_act_cri` and `mini_batch_generator` on the deterministic
inputs of `cases.py`.  Only inputs/outputs (data) are written; no reference
source travels.  Run:  python -m tests.golden.make_golden   (from the repo root)
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    utils = types.ModuleType("utils")
    utils.__path__ = []
    utils.path2video = lambda *a, **k: None
    sys.modules["utils"] = utils
    sys.modules["utils.torch_jit_utils"] = types.ModuleType("utils.torch_jit_utils")
    sys.modules["torchvision"] = types.ModuleType("torchvision")
    # our own top-level `algorithms` shim must not shadow the reference's package
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
    sys.path.insert(0, REF)
    for m in [m for m in sys.modules if m == "algorithms" or m.startswith("algorithms.")]:
        del sys.modules[m]
    import algorithms  # noqa
    assert algorithms.__file__.startswith(REF), algorithms.__file__
    return algorithms


class FakeEnv:
    def __init__(self, num_envs, num_obs, num_actions):
        self.num_envs, self.num_obs, self.num_actions = num_envs, num_obs, num_actions
        self.max_episode_length = 200


class FakeLogger:
    def __init__(self, d):
        self.save_ckpt_dir = d


def ppo_cfg(c, num_envs, obs_mode="normal_state"):
    return dict(num_envs=num_envs, obs_mode=obs_mode, succ_value=c.get("succ_value"),
                model=dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0,
                           network=dict(c["net"])),
                max_iterations=c["max_iterations"], n_steps=c["T"], n_updates=c["n_updates"],
                n_minibatches=c["n_minibatches"], device="cpu", eval_round=1, eval_frequence=10 ** 9,
                save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
                lr_schedule=c["lr_schedule"], lr=c["lr"], desired_kl=c["desired_kl"],
                epsilon_clip=c["epsilon_clip"], gamma=c["gamma"], lam=c["lam"], tricks=dict(c["tricks"]),
                sampler=c["sampler"], resume=None)


def load_sd(module, sd_np):
    module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})


def flat_params(sd):
    return np.concatenate([v.detach().cpu().numpy().reshape(-1).astype(np.float32) for v in sd.values()])


class Trace:
    """Records every `loss.backward()` value and a float64 checksum after every optimiser step."""

    def __init__(self, params_fn):
        self.losses, self.sums = [], []
        self.params_fn = params_fn

    def __enter__(self):
        self._bw = torch.Tensor.backward
        tr = self

        def bw(t, *a, **k):
            tr.losses.append(float(t.detach()))
            return tr._bw(t, *a, **k)
        torch.Tensor.backward = bw
        self._step = torch.optim.Adam.step

        def step(opt, *a, **k):
            r = tr._step(opt, *a, **k)
            ps = tr.params_fn()
            tr.sums.append([float(sum(p.detach().double().sum() for p in ps)),
                            float(sum((p.detach().double() ** 2).sum() for p in ps))])
            return r
        torch.optim.Adam.step = step
        return self

    def __exit__(self, *a):
        torch.Tensor.backward = self._bw
        torch.optim.Adam.step = self._step


def gen_gae(ref_algos, cases):
    from algorithms.algo_utils import RolloutStorage
    for name, c in cases.GAE_CASES.items():
        inp = cases.gae_inputs(c)
        T, N = c["T"], c["N"]
        st = RolloutStorage(N, T, 3, 2, "cpu", c["succ_value"], c["whole_adv_norm"])
        st.rewards.copy_(torch.from_numpy(inp["rewards"]))
        st.values.copy_(torch.from_numpy(inp["values"]))
        st.dones.copy_(torch.from_numpy(inp["dones"]))
        st.succs.copy_(torch.from_numpy(inp["succs"]))
        st.compute_returns(torch.from_numpy(inp["last_values"]), 0.99, 0.95)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), returns=st.returns.numpy(),
                            advantages=st.advantages.numpy())
        print("wrote", name)


def gen_ppo(ref_algos, cases):
    from algorithms import ppo
    for name, c in cases.PPO_CASES.items():
        T, N, O, A = c["T"], c["N"], c["O"], c["A"]
        env = FakeEnv(N, {"normal_state": O}, A)
        with tempfile.TemporaryDirectory() as d:
            run = ppo(env, ppo_cfg(c, N), FakeLogger(d))
        sd0 = cases.actor_critic_state(c["net"], O, A, c["action_std"], c["seed"])
        load_sd(run.actor_critic, sd0)
        raw = cases.ppo_raw_inputs(c)
        obs = torch.from_numpy(raw["observations"])
        act = torch.from_numpy(raw["actions"])
        with torch.no_grad():
            logp, ent, val, mu, sig = run.actor_critic.update_act_cri(obs.view(-1, O).clone(), act.view(-1, A))
        nz = raw["noise"]
        on = c["old_noise"] * (torch.arange(1, T + 1).view(T, 1, 1) / T if c["noise_ramp"] else torch.ones(T, 1, 1))
        values = val.view(T, N, 1) + 0.5 * torch.from_numpy(nz["values"])
        old_logp = logp.view(T, N, 1) + on * torch.from_numpy(nz["logp"])
        old_mu = mu.view(T, N, A) + on * torch.from_numpy(nz["mu"])
        old_sigma = sig.view(T, N, A) + 0.02 * torch.from_numpy(nz["sigma"])
        with torch.no_grad():
            last_values = run.actor_critic.cri(obs[-1].clone()) + 0.5 * torch.from_numpy(nz["last_values"])
        for t in range(T):
            run.storage.add_transitions(obs[t], act[t], torch.from_numpy(raw["rewards"][t, :, 0]),
                                        torch.from_numpy(raw["dones"][t, :, 0]),
                                        torch.from_numpy(raw["succs"][t, :, 0]),
                                        values[t], old_logp[t, :, 0], old_mu[t], old_sigma[t])
        run.storage.compute_returns(last_values, c["gamma"], c["lam"])
        out = dict(values=values.numpy(), actions_log_prob=old_logp.numpy(), mu=old_mu.numpy(),
                   sigma=old_sigma.numpy(), last_values=last_values.numpy(),
                   fwd_logp=logp.numpy(), fwd_entropy=ent.numpy(), fwd_value=val.numpy(), fwd_mu=mu.numpy(),
                   returns=run.storage.returns.numpy().copy(),
                   advantages=run.storage.advantages.numpy().copy())
        if c["sampler"] == "random":
            torch.manual_seed(c["seed"])
            gen = run.storage.mini_batch_generator(c["n_minibatches"])
            out["index_lists"] = np.array([[list(b) for b in gen] for _ in range(2 * c["n_updates"])], dtype=np.int64)
            torch.manual_seed(c["seed"])
        else:
            gen = run.storage.mini_batch_generator(c["n_minibatches"])
            out["index_lists"] = np.array([[list(b) for b in gen]], dtype=np.int64)
        run.log_dict = {}
        with Trace(lambda: list(run.actor_critic.parameters())) as tr:
            run.update(c["it"])
        out["loss_trace"] = np.array(tr.losses, dtype=np.float64)
        out["sum_trace"] = np.array(tr.sums, dtype=np.float64)
        for k in ("value_function_loss", "surrogate_loss", "kl", "kl_max", "kl_update_count", "learning_rate",
                  "value_gt_return_mean", "value_gt_return_max"):
            out["log_" + k] = np.float64(float(run.log_dict["Train/" + k]))
        out["lr_actor_groups"] = np.array([g["lr"] for g in run.optimizer_actor.param_groups])
        out["lr_critic_groups"] = np.array([g["lr"] for g in run.optimizer_critic.param_groups])
        fin = flat_params(run.actor_critic.state_dict())
        stride = 3 if c["net"]["name"] == "PointNet" else 1
        out["final_flat"] = fin[::stride]
        out["final_stride"] = np.int64(stride)
        out["final_sum"] = np.float64(fin.astype(np.float64).sum())
        out["final_sq"] = np.float64((fin.astype(np.float64) ** 2).sum())
        # Adam moments of the policy head (small) pin the optimiser state too
        st_a = run.optimizer_actor.state_dict()["state"]
        last = max(k for k in st_a.keys() if st_a[k]["exp_avg"].numel() > 0)
        out["adam_logstd_m"] = st_a[last]["exp_avg"].numpy()
        out["adam_logstd_v"] = st_a[last]["exp_avg_sq"].numpy()
        out["adam_step"] = np.float64(float(st_a[last]["step"]))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, "losses", len(tr.losses), "count", out["log_kl_update_count"])
        if name == "ppo_mlp_default":
            # a checkpoint written by the REFERENCE's own ppo.save (ppo.py:83-100): wire-format fixture for
            # resume() / FusedAdam.load_state_dict (data only: tensors + plain dicts)
            with tempfile.TemporaryDirectory() as d:
                run.save_ckpt_dir = d
                run.total_envsteps = 12345
                run.save(c["it"])
                import shutil
                shutil.copy(os.path.join(d, f"model_{c['it']}.pth"), os.path.join(HERE, "ref_ckpt_ppo_mlp_default.pth"))
            print("wrote ref_ckpt_ppo_mlp_default.pth")


def gen_dagger(ref_algos, cases):
    from algorithms import ppo, dagger
    for name, c in cases.DAGGER_CASES.items():
        N, A = c["N"], c["A"]
        with tempfile.TemporaryDirectory() as d:
            cwd = os.getcwd()
            os.chdir(d)
            try:
                np.save("teacher_reward.npy", np.linspace(0, 1, 200).astype(np.float32))
                tc = dict(net=c["tea_net"], T=1, n_updates=1, n_minibatches=1, tricks=dict(cases.TRICKS_DEFAULT),
                          sampler="sequential", succ_value=None, lr=1e-3, desired_kl=0.1, lr_schedule="fixed",
                          gamma=0.99, lam=0.95, epsilon_clip=0.2, action_std=0.5, max_iterations=10)
                env_t = FakeEnv(N, {"normal_state": c["O_t"]}, A)
                tea_run = ppo(env_t, ppo_cfg(tc, N), FakeLogger(d))
                load_sd(tea_run.actor_critic, cases.actor_critic_state(c["tea_net"], c["O_t"], A, 0.5, c["seed"] + 1))
                tea_run.save(1)
                env = FakeEnv(N, {"stu_mode": c["O_s"], "normal_state": c["O_t"], "proprio_state": c["proprio"]}, A)
                cfg = dict(num_envs=N, obs_mode="stu_mode",
                           model=dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0,
                                      network=dict(c["stu_net"])),
                           max_iterations=c["max_iterations"], n_steps=1, n_updates=c["n_updates"],
                           n_minibatches=c["n_minibatches"], device="cpu", buf_size=c["buf_size"],
                           reward_reset=True, add_proprio_obs=c["proprio"] > 0, offline_data_pth=None,
                           eval_round=1, eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False,
                           save_pose=False, save_video=False, lr_schedule=c["lr_schedule"], lr=c["lr"],
                           teacher=os.path.join(d, "model_1.pth"), resume=None, pretrain=None,
                           sampler=c["sampler"])
                run = dagger(env, cfg, FakeLogger(d))
            finally:
                os.chdir(cwd)
        load_sd(run.student, cases.actor_critic_state(c["stu_net"], c["O_s"], A, c["action_std"], c["seed"],
                                                      c["proprio"]))
        raw = cases.dagger_raw_inputs(c)
        for k in range(c["n_fill"]):
            run.storage.add_transitions_dagger(torch.from_numpy(raw["stu"][k]), torch.from_numpy(raw["tea"][k]))
        out = dict(ring_obs_sum=np.float64(run.storage.observations.double().sum()),
                   ring_tea=run.storage.tea_obs.numpy().copy(), mix_buf_ind=np.int64(run.storage.mix_buf_ind),
                   cur_buf_size=np.int64(run.storage.cur_buf_size))
        with torch.no_grad():
            out["tea_act"] = run.teacher.act(run.storage.tea_obs).numpy()
            out["stu_act0"] = run.student.act(run.storage.observations.clone()).numpy()
        torch.manual_seed(c["torch_seed"])
        lists = []
        for _ in range(c["n_updates"]):
            lists.append([list(b) for b in run.storage.mini_batch_generator(c["n_minibatches"])])
        out["index_lists"] = np.array(lists, dtype=np.int64)
        torch.manual_seed(c["torch_seed"])
        run.log_dict = {}
        with Trace(lambda: list(run.student.parameters())) as tr:
            run.update(c["it"])
        out["loss_trace"] = np.array(tr.losses, dtype=np.float64)
        out["sum_trace"] = np.array(tr.sums, dtype=np.float64)
        out["log_dagger_loss"] = np.float64(float(run.log_dict["Train/dagger_loss"]))
        out["log_learning_rate"] = np.float64(float(run.log_dict["Train/learning_rate"]))
        fin = flat_params(run.student.state_dict())
        stride = 3 if c["stu_net"]["name"] in ("PointNet", "Conv3DNet") else 1
        out["final_flat"] = fin[::stride]
        out["final_stride"] = np.int64(stride)
        out["final_sum"] = np.float64(fin.astype(np.float64).sum())
        out["final_sq"] = np.float64((fin.astype(np.float64) ** 2).sum())
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, "losses", out["loss_trace"])


def gen_dagger_ckpt(ref_algos, cases):
    """A student checkpoint written by the REFERENCE's own `dagger.save` (dagger.py:81-95) after one update, then -- by the
    reference again -- `dagger(..., resume=ckpt)` (dagger.py:107-120) continuing with a second update on the same ring, and
    `load_pretrain(ckpt)` (dagger.py:98-105) into a differently initialised student.  Data only: tensors + plain dicts."""
    import shutil
    from algorithms import ppo, dagger
    c = cases.DAGGER_CASES["dagger_mlp"]
    N, A = c["N"], c["A"]
    raw = cases.dagger_raw_inputs(c)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        cwd = os.getcwd()
        os.chdir(d)
        try:
            np.save("teacher_reward.npy", np.linspace(0, 1, 200).astype(np.float32))
            tc = dict(net=c["tea_net"], T=1, n_updates=1, n_minibatches=1, tricks=dict(cases.TRICKS_DEFAULT),
                      sampler="sequential", succ_value=None, lr=1e-3, desired_kl=0.1, lr_schedule="fixed",
                      gamma=0.99, lam=0.95, epsilon_clip=0.2, action_std=0.5, max_iterations=10)
            tea_run = ppo(FakeEnv(N, {"normal_state": c["O_t"]}, A), ppo_cfg(tc, N), FakeLogger(d))
            load_sd(tea_run.actor_critic, cases.actor_critic_state(c["tea_net"], c["O_t"], A, 0.5, c["seed"] + 1))
            tea_run.save(1)
            env = FakeEnv(N, {"stu_mode": c["O_s"], "normal_state": c["O_t"], "proprio_state": c["proprio"]}, A)

            def make(resume=None, pretrain=None):
                cfg = dict(num_envs=N, obs_mode="stu_mode",
                           model=dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0, network=dict(c["stu_net"])),
                           max_iterations=c["max_iterations"], n_steps=1, n_updates=c["n_updates"], n_minibatches=c["n_minibatches"],
                           device="cpu", buf_size=c["buf_size"], reward_reset=True, add_proprio_obs=False, offline_data_pth=None,
                           eval_round=1, eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False,
                           save_video=False, lr_schedule=c["lr_schedule"], lr=c["lr"], teacher=os.path.join(d, "model_1.pth"),
                           resume=resume, pretrain=pretrain, sampler=c["sampler"])
                run = dagger(env, cfg, FakeLogger(d))
                for k in range(c["n_fill"]):
                    run.storage.add_transitions_dagger(torch.from_numpy(raw["stu"][k]), torch.from_numpy(raw["tea"][k]))
                run.log_dict = {}
                return run

            run = make()
            load_sd(run.student, cases.actor_critic_state(c["stu_net"], c["O_s"], A, c["action_std"], c["seed"], c["proprio"]))
            torch.manual_seed(c["torch_seed"])
            run.update(c["it"])
            run.total_envsteps = 4321
            os.makedirs("stu", exist_ok=True)
            run.save_ckpt_dir = os.path.join(d, "stu")
            run.save(c["it"])
            ck = os.path.join(d, "stu", f"model_{c['it']}.pth")
            shutil.copy(ck, os.path.join(HERE, "ref_ckpt_dagger_mlp.pth"))
            out["saved_flat"] = flat_params(run.student.state_dict())
            # ---- resume, by the reference
            run2 = make(resume=ck)
            assert run2.curr_iter == c["it"] and run2.total_envsteps == 4321
            assert np.array_equal(flat_params(run2.student.state_dict()), out["saved_flat"])
            torch.manual_seed(c["torch_seed"] + 1)
            with Trace(lambda: list(run2.student.parameters())) as tr:
                run2.update(c["it"] + 1)
            out["resume_loss_trace"] = np.array(tr.losses, dtype=np.float64)
            out["resume_log_dagger_loss"] = np.float64(float(run2.log_dict["Train/dagger_loss"]))
            out["resume_log_learning_rate"] = np.float64(float(run2.log_dict["Train/learning_rate"]))
            out["resume_final_flat"] = flat_params(run2.student.state_dict())
            st = run2.optimizer.state_dict()["state"]
            out["resume_adam_steps"] = np.array([float(st[k]["step"]) for k in sorted(st)], dtype=np.float64)
            # ---- load_pretrain, by the reference: every tensor but log_std comes from the checkpoint
            run3 = make()
            other = cases.actor_critic_state(c["stu_net"], c["O_s"], A, 0.3, c["seed"] + 50, c["proprio"])
            load_sd(run3.student, other)
            run3.load_pretrain(ck)
            out["pretrain_flat"] = flat_params(run3.student.state_dict())
            out["pretrain_log_std"] = run3.student.log_std.detach().numpy().copy()
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "dagger_mlp_ckpt.npz"), **out)
    print("wrote dagger_mlp_ckpt + ref_ckpt_dagger_mlp.pth; resume losses", out["resume_loss_trace"], "adam steps", out["resume_adam_steps"][:3])


def gen_dagger_offline(ref_algos, cases):
    """Mixed BC + on-policy DAgger by the reference's own code: `RolloutStorage.add_transitions_offline`
    (storage.py:58-82, the call `dagger.run` makes first, dagger.py:186-187) on shards written to disk, then
    `add_transitions_dagger` for the on-policy steps (ring wraps over the offline rows), then `dagger.update`."""
    from algorithms import ppo, dagger
    c = cases.DAGGER_OFFLINE_CASE
    N, A, O_s = c["N"], c["A"], c["D"] + c["proprio"]
    with tempfile.TemporaryDirectory() as d:
        cwd = os.getcwd()
        os.chdir(d)
        try:
            np.save("teacher_reward.npy", np.linspace(0, 1, 200).astype(np.float32))
            cases.dagger_offline_write(c, os.path.join(d, "offline"))
            tc = dict(net=c["tea_net"], T=1, n_updates=1, n_minibatches=1, tricks=dict(cases.TRICKS_DEFAULT),
                      sampler="sequential", succ_value=None, lr=1e-3, desired_kl=0.1, lr_schedule="fixed",
                      gamma=0.99, lam=0.95, epsilon_clip=0.2, action_std=0.5, max_iterations=10)
            tea_run = ppo(FakeEnv(N, {"normal_state": c["O_t"]}, A), ppo_cfg(tc, N), FakeLogger(d))
            load_sd(tea_run.actor_critic, cases.actor_critic_state(c["tea_net"], c["O_t"], A, 0.5, c["seed"] + 1))
            tea_run.save(1)
            env = FakeEnv(N, {"tsdf": O_s, "normal_state": c["O_t"], "proprio_state": c["proprio"]}, A)
            cfg = dict(num_envs=N, obs_mode="tsdf",
                       model=dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0, network=dict(c["stu_net"])),
                       max_iterations=c["max_iterations"], n_steps=1, n_updates=c["n_updates"], n_minibatches=c["n_minibatches"],
                       device="cpu", buf_size=c["buf_size"], reward_reset=True, add_proprio_obs=True,
                       offline_data_pth=os.path.join(d, "offline"), eval_round=1, eval_frequence=10 ** 9,
                       save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
                       lr_schedule=c["lr_schedule"], lr=c["lr"], teacher=os.path.join(d, "model_1.pth"), resume=None,
                       pretrain=None, sampler=c["sampler"])
            run = dagger(env, cfg, FakeLogger(d))
            # the student's first Linear takes [tsdf | proprio]: the reference sizes it from stu_input_obs = num_obs['tsdf']
            load_sd(run.student, cases.actor_critic_state(c["stu_net"], O_s, A, c["action_std"], c["seed"]))
            run.storage.add_transitions_offline(run.offline_data_pth, run.device, run.add_proprio_obs)   # dagger.py:186-187
        finally:
            os.chdir(cwd)
    st = run.storage
    out = dict(off_ring_obs=st.observations.numpy().copy(), off_ring_tea=st.tea_obs.numpy().copy(),
               off_state=np.array([st.mix_buf_ind, st.cur_buf_size, st.last_episode_buf_ind], dtype=np.int64))
    on = cases.dagger_offline_online(c)
    for k in range(c["n_fill"]):
        st.add_transitions_dagger(torch.from_numpy(on["stu"][k]), torch.from_numpy(on["tea"][k]))
    out.update(ring_obs=st.observations.numpy().copy(), ring_tea=st.tea_obs.numpy().copy(),
               state=np.array([st.mix_buf_ind, st.cur_buf_size, st.last_episode_buf_ind], dtype=np.int64))
    torch.manual_seed(c["torch_seed"])
    run.log_dict = {}
    with Trace(lambda: list(run.student.parameters())) as tr:
        run.update(c["it"])
    out["loss_trace"] = np.array(tr.losses, dtype=np.float64)
    out["log_dagger_loss"] = np.float64(float(run.log_dict["Train/dagger_loss"]))
    out["log_learning_rate"] = np.float64(float(run.log_dict["Train/learning_rate"]))
    out["final_flat"] = flat_params(run.student.state_dict())
    np.savez_compressed(os.path.join(HERE, "dagger_offline.npz"), **out)
    print("wrote dagger_offline", out["off_state"], out["state"], "losses", out["loss_trace"])


def gen_bc(ref_algos, cases):
    """The reference's own bc.run() (10 DataLoader workers) on shards written by cases.bc_write_dataset."""
    from algorithms import bc
    for name, c in cases.BC_CASES.items():
        with tempfile.TemporaryDirectory() as d:
            cases.bc_write_dataset(c, os.path.join(d, "data"))
            env = FakeEnv(4, {"tsdf": c["D"] + c["S"], "proprio_state": c["S"]}, c["A"])
            trace = []

            class Log(FakeLogger):
                def info(self, log, it):
                    trace.append((float(log["Train/bc_loss"]), float(log["Train/learning_rate"])))

            cfg = dict(num_envs=4, obs_mode="tsdf", model=dict(action_std=c["action_std"], action_activate="tanh",
                                                                clipAction=1.0, network=dict(c["net"])),
                       max_iterations=c["max_iterations"], device="cpu", data_path=os.path.join(d, "data"),
                       n_minibatches=c["n_minibatches"], add_proprio_obs=True, eval_round=1, eval_frequence=10 ** 9,
                       save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
                       lr_schedule=c["lr_schedule"], lr=c["lr"], resume=None)
            run = bc(env, cfg, Log(d))
            sd0 = cases.actor_critic_state(c["net"], c["D"] + c["S"], c["A"], c["action_std"], c["seed"])
            load_sd(run.student, sd0)
            torch.manual_seed(c["torch_seed"])
            run.run()
            out = dict(loss_trace=np.asarray([t[0] for t in trace], dtype=np.float64),
                       lr_trace=np.asarray([t[1] for t in trace], dtype=np.float64),
                       final_flat=flat_params(run.student.state_dict()))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, "losses", out["loss_trace"])


def gen_conv3d(ref_algos, cases):
    """The reference's own Conv3DNet (algorithms/algo_utils/network.py:67-94): forward outputs and a strided sample
    of every parameter gradient of sum(out * dy)."""
    from algorithms.algo_utils.network import Conv3DNet
    for name, c in cases.CONV3D_CASES.items():
        net = Conv3DNet(c["res"] ** 3, c["out"], {"activation": "tanh"}, c["proprio"])
        load_sd(net, cases.conv3d_state(c))
        inp = cases.conv3d_inputs(c)
        out = net(torch.from_numpy(inp["x"]))
        (out * torch.from_numpy(inp["dy"])).sum().backward()
        fx = dict(out=out.detach().numpy())
        for k, v in net.named_parameters():
            fx["grad_" + k] = v.grad.numpy().reshape(-1)[::7].copy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
        print("wrote", name, "out[0]", fx["out"][0][:3])


def gen_depth2pc(cases):
    """Runs the reference's own TSDFVolume.depth2pc (utils/depth2tsdf.py) file by path.  Its two absent
    dependencies are stubbed: `skimage` (unused by this method) and `pytorch3d.ops.sample_farthest_points`,
    which is replaced by the CPU restatement's fps and RECORDS its input -- so `world` (everything before the
    sampling) is the reference's own output, and `final_pc` = reference gather of restated indices."""
    import importlib.util
    sys.modules["skimage"] = types.ModuleType("skimage")
    sys.modules["skimage"].measure = types.ModuleType("skimage.measure")
    sys.modules["skimage.measure"] = sys.modules["skimage"].measure
    from oracle import ref_cpu as R
    rec = {}

    def sample_farthest_points(points, K):
        rec["points"] = points.detach().cpu().numpy().copy()
        idx = torch.from_numpy(R.fps(rec["points"].astype(np.float32), K))
        got = torch.gather(points, 1, idx.clamp(min=0).unsqueeze(-1).expand(-1, -1, 3))
        return got * (idx >= 0).unsqueeze(-1), idx                # pytorch3d's masked_gather: -1 -> zero row

    p3d, p3d_ops = types.ModuleType("pytorch3d"), types.ModuleType("pytorch3d.ops")
    p3d_ops.sample_farthest_points = sample_farthest_points
    p3d.ops = p3d_ops
    sys.modules["pytorch3d"], sys.modules["pytorch3d.ops"] = p3d, p3d_ops
    spec = importlib.util.spec_from_file_location("ref_depth2tsdf", os.path.join(REF, "utils", "depth2tsdf.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, c in cases.DEPTH2PC_CASES.items():
        inp = cases.depth2pc_inputs(c)
        vol = mod.TSDFVolume("cpu", size=c["size"], resolution=10, _vol_origin=c["vol_origin"])
        vol.register_camera(inp["cam_pose"], np.asarray(c["intr"], dtype=np.float32), c["h"], c["w"], c["b"])
        final = vol.depth2pc(torch.from_numpy(inp["depth"]))
        rec["world"] = rec["points"]
        sparse = vol.sparse_voxel(torch.from_numpy(inp["depth"])).numpy().copy()      # depth2tsdf.py:88-120, K = 1024
        tsdf = vol.integrate(torch.from_numpy(inp["depth"])).numpy().copy()          # depth2tsdf.py:68-86, resolution 10
        # the method hard-codes K=1024 (depth2tsdf.py:160); the case's K-sample prefix is what the tests compare
        idx = R.fps(rec["world"], c["K"])
        # registration-time voxel -> pixel tables of the reference (they come out of a host bmm whose last bit
        # depends on the CPU, so they are part of the fixture)
        pix_idx = torch.where(vol.valid_pix, vol.valid_pix_y * c["w"] + vol.valid_pix_x,
                              torch.full_like(vol.valid_pix_x, -1)).numpy().astype(np.int32)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), world=rec["world"], final_pc_1024=final.numpy(),
                            idx=idx.astype(np.int32), tsdf=tsdf, tsdf_pix_idx=pix_idx, tsdf_pix_z=vol.pix_z.numpy(),
                            sparse_voxel=sparse)
        print("wrote", name, "valid fraction", float((rec["world"] != 0).any(-1).mean()),
              "band voxels", ((tsdf < 0.2) & (tsdf > -0.2)).reshape(tsdf.shape[0], -1).sum(-1))


def gen_rollout(ref_algos, cases):
    """Rollout side, by the reference's own classes: `Normalization` (RMS.py:36-45) fed three
    observation batches, and `ActorCritic.random_act_cri` (actor_critic.py:36-47) under a fixed torch seed -- with the
    standard-normal draw `MultivariateNormal.sample` consumes recorded next to its outputs."""
    from algorithms.algo_utils import Normalization, ActorCritic
    from tests.golden.detgen import det_uniform, det_normal
    c = cases.ROLLOUT_CASE
    N, O, A = c["N"], c["O"], c["A"]
    norm = Normalization(O, "cpu")
    outs, stats = [], []
    for i in range(3):
        x = torch.from_numpy((det_normal((N, O), c["seed"] + i) * (1.0 + 0.5 * i) + det_uniform((1, O), c["seed"] + 9, -2, 2))
                             .astype(np.float32))
        outs.append(norm(x).numpy().copy())
        stats.append(np.stack([norm.running_ms.mean.numpy()[0], norm.running_ms.std.numpy()[0], norm.running_ms.S.numpy()[0]]))
    frozen = norm(x, update=False).numpy().copy()                     # Normalization(x, update=False): eval path
    ac = ActorCritic(O, A, dict(action_std=c["action_std"], action_activate="tanh", clipAction=c["max_action"],
                                network=c["net"]))
    load_sd(ac, cases.actor_critic_state(c["net"], O, A, c["action_std"], c["seed"]))
    obs = torch.from_numpy(det_normal((N, O), c["seed"] + 20).astype(np.float32))
    torch.manual_seed(c["torch_seed"])
    with torch.no_grad():
        act, logp, val, mu, ls = ac.random_act_cri(obs)
    torch.manual_seed(c["torch_seed"])
    eps = torch.normal(torch.zeros(N, A), torch.ones(N, A))
    np.savez_compressed(os.path.join(HERE, "rollout_side.npz"), norm_out=np.stack(outs), norm_stats=np.stack(stats),
                        norm_frozen=frozen, n=np.int64(norm.running_ms.n), actions=act.numpy(), logp=logp.numpy(),
                        value=val.numpy(), mu=mu.numpy(), log_std_rows=ls.numpy(), eps=eps.numpy())
    print("wrote rollout_side", "n", norm.running_ms.n, "std range", float(stats[-1][1].min()), float(stats[-1][1].max()))


def main():
    torch.set_num_threads(8)
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
    from tests.golden import cases
    which = sys.argv[1:] or ["gae", "ppo", "dagger", "dagger_ckpt", "dagger_offline", "depth2pc", "bc", "conv3d", "rollout"]
    if "depth2pc" in which:
        gen_depth2pc(cases)
        which = [w for w in which if w != "depth2pc"]
        if not which:
            return
    ref = _import_reference()
    if "gae" in which:
        gen_gae(ref, cases)
    if "ppo" in which:
        gen_ppo(ref, cases)
    if "dagger" in which:
        gen_dagger(ref, cases)
    if "dagger_ckpt" in which:
        gen_dagger_ckpt(ref, cases)
    if "dagger_offline" in which:
        gen_dagger_offline(ref, cases)
    if "bc" in which:
        gen_bc(ref, cases)
    if "conv3d" in which:
        gen_conv3d(ref, cases)
    if "rollout" in which:
        gen_rollout(ref, cases)


if __name__ == "__main__":
    main()
