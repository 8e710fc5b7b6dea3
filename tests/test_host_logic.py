"""Host-side logic that needs no GPU: storage bookkeeping, sampler semantics, ring buffer,
checkpoint-format plumbing of FusedAdam, feeder duck-type, config loading."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.golden import cases
from tests.helpers import load_fixture


def test_storage_sampler_matches_reference_index_lists():
    from partmanip_amd.algo_utils import RolloutStorage
    for name, c in cases.PPO_CASES.items():
        fx = load_fixture(name)
        st = RolloutStorage(c["N"], c["T"], 4, 2, "cpu", sampler=c["sampler"])
        gen = st.mini_batch_generator(c["n_minibatches"])
        if c["sampler"] == "random":
            torch.manual_seed(c["seed"])
            got = [[list(b) for b in gen] for _ in range(2 * c["n_updates"])]
        else:
            got = [[list(b) for b in gen]]
        assert np.array_equal(np.array(got, dtype=np.int64), fx["index_lists"]), name


def test_minibatch_cap_2048_and_drop_last():
    from partmanip_amd.algo_utils import RolloutStorage
    st = RolloutStorage(4096, 8, 1, 1, "cpu")
    b = st.mini_batch_generator(8)
    assert len(b) == 16 and all(len(x) == 2048 for x in b)
    st = RolloutStorage(7, 9, 1, 1, "cpu")
    b = list(st.mini_batch_generator(4))
    assert len(b) == 4 and all(len(x) == 15 for x in b) and b[-1][-1] == 59


def test_storage_overflow_and_clear():
    from partmanip_amd.algo_utils import RolloutStorage
    st = RolloutStorage(3, 2, 5, 2, "cpu")
    z = torch.zeros
    args = (z(3, 5), z(3, 2), z(3), z(3, dtype=torch.bool), z(3, dtype=torch.bool), z(3, 1), z(3), z(3, 2), z(3, 2))
    st.add_transitions(*args)
    st.add_transitions(*args)
    with pytest.raises(AssertionError, match="Rollout buffer overflow"):
        st.add_transitions(*args)
    st.clear()
    st.add_transitions(*args)
    assert st.dones.dtype == torch.bool and st.observations.shape == (2, 3, 5)


def test_dagger_ring_wraps_like_oracle():
    from partmanip_amd.algo_utils import RolloutStorage
    c = cases.DAGGER_CASES["dagger_mlp"]
    raw = cases.dagger_raw_inputs(c)
    st = RolloutStorage(c["N"], c["buf_size"], c["O_s"], c["A"], "cpu", sampler="random", tea_obs_shape=c["O_t"], max_length=200)
    cap = c["buf_size"] * c["N"]
    ro, rt, ind, size = torch.zeros(cap, c["O_s"]), torch.zeros(cap, c["O_t"]), 0, 0
    for k in range(c["n_fill"]):
        st.add_transitions_dagger(torch.from_numpy(raw["stu"][k]), torch.from_numpy(raw["tea"][k]))
        ind, size = R.dagger_ring_insert(ro, rt, ind, size, torch.from_numpy(raw["stu"][k]), torch.from_numpy(raw["tea"][k]))
    assert (st.mix_buf_ind, st.cur_buf_size) == (ind, size)
    assert torch.equal(st.observations, ro) and torch.equal(st.tea_obs, rt)


def test_compute_returns_refuses_cpu():
    from partmanip_amd.algo_utils import RolloutStorage
    st = RolloutStorage(3, 2, 5, 2, "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        st.compute_returns(torch.zeros(3, 1), 0.99, 0.95)


def test_running_mean_std_matches_oracle():
    from partmanip_amd.algo_utils import Normalization
    n = Normalization(5, "cpu")
    r = R.RunningMeanStd(5)
    g = torch.Generator().manual_seed(0)
    for _ in range(4):
        x = torch.randn(17, 5, generator=g) * 3 + 1
        y = n(x, update=True)
        r.update(x)
        assert torch.equal(n.running_ms.mean, r.mean) and torch.equal(n.running_ms.std, r.std)
        assert torch.equal(y, (x - r.mean) / r.std)
    d = n.running_ms.save()
    assert set(d) == {"mean", "std", "S", "n"} and d["n"] == 4


def test_state_dict_keys_match_reference_layout():
    from partmanip_amd.algo_utils import ActorCritic
    for c in (cases.PPO_CASES["ppo_mlp_default"], cases.PPO_CASES["ppo_pn_maxmean"]):
        ac = ActorCritic(c["O"], c["A"], dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=dict(c["net"])))
        want = cases.actor_critic_state(c["net"], c["O"], c["A"], 0.5, 1)
        got = ac.state_dict()
        assert list(got.keys()) == list(want.keys())
        for k in want:
            assert tuple(got[k].shape) == want[k].shape, k
    # policy-head orthogonal init gains (network.py:44-51): actor head tiny, critic head unit-norm rows
    c = cases.PPO_CASES["ppo_mlp_default"]
    ac = ActorCritic(c["O"], c["A"], dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=dict(c["net"])))
    assert float(ac.actor.model[6].weight.norm(dim=1).max()) < 0.011
    np.testing.assert_allclose(float(ac.critic.model[6].weight.norm()), 1.0, rtol=1e-4)
    np.testing.assert_allclose(ac.log_std.detach().numpy(), np.log(0.5), rtol=1e-6)


def test_process_cfgs_merges_and_overrides(tmp_path):
    import os
    from partmanip_amd.config import process_cfgs, num_actions
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = process_cfgs(["--algocfg", "ppo_pointnet", "--taskcfg", "open_drawer", "--algo.lr", "1e-4",
                        "--algo.tricks.use_grad_clip", "--algo.model.network.max_mean", "--exp_name", "t"], root=root)
    assert cfg["algo_name"] == "ppo" and cfg["task_name"] == "open_drawer"
    assert cfg["algo"]["lr"] == 1e-4
    assert cfg["algo"]["tricks"]["use_grad_clip"] is False          # bool leaves are toggles (utils/config.py:56-60)
    assert cfg["algo"]["model"]["network"]["max_mean"] is False
    assert cfg["algo"]["model"]["clipAction"] == 1.0 and cfg["algo"]["succ_value"] is None
    assert cfg["algo"]["device"] == "cuda:0" and cfg["task"]["num_envs"] == 4096
    assert cfg["task"]["obs_mode"]["depth_pc"] == 3072 and num_actions(cfg["task"]) == 10
    cfg = process_cfgs(["--taskcfg", "grasp_cube"], root=root)
    assert cfg["algo"]["succ_value"] == 500 and num_actions(cfg["task"]) == 7


def test_feeder_env_duck_type_cpu():
    from partmanip_amd.feeder import FeederEnv
    env = FeederEnv(6, {"normal_state": 53, "depth_pc": 3072, "proprio_state": 0}, 10, "cpu", seed=3)
    obs = env.reset()
    assert obs["normal_state"].shape == (6, 53) and obs["depth_pc"].shape == (6, 3072)
    o2, rew, done, extras = env.step(torch.zeros(6, 10))
    assert rew.shape == (6,) and done.dtype == torch.bool and extras["succ_rate"].shape == (1,)
    assert env.reset_succ.shape == (6,) and (env.reset_succ <= done).all()
    assert float(o2["depth_pc"].abs().max()) <= 1.5
