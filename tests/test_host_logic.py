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


def test_reference_checkpoint_wire_format_loads():
    """A checkpoint written by the REFERENCE's ppo.save (tests/golden/ref_ckpt_ppo_mlp_default.pth) loads into
    this build's ActorCritic / FusedAdam (same state_dict keys, torch.optim.Adam state layout), and what we
    write back has the same structure."""
    import os
    from partmanip_amd.algo_utils import ActorCritic, FusedAdam
    from tests.helpers import GOLDEN, flat_state
    c, fx = cases.PPO_CASES["ppo_mlp_default"], load_fixture("ppo_mlp_default")
    ck = torch.load(os.path.join(GOLDEN, "ref_ckpt_ppo_mlp_default.pth"), map_location="cpu", weights_only=False)
    assert ck["iteration"] == c["it"] and ck["total_steps"] == 12345 and ck["obs_mode"] == "normal_state"
    ac = ActorCritic(c["O"], c["A"], ck["model_cfg"])
    f = ac.flat()
    ac.load_state_dict(ck["model_state_dict"])
    np.testing.assert_array_equal(flat_state(ac.state_dict()), fx["final_flat"])
    n_a, A = f["n_actor"], c["A"]
    assert torch.equal(f["actor"][n_a:], ac.log_std.data)            # log_std lives at the tail of the actor buffer
    opt_a = FusedAdam(f["actor"], f["grad_actor"][:n_a + A], [list(ac.actor.parameters()), [ac.log_std]], lr=1.0)
    opt_c = FusedAdam(f["critic"], f["grad_critic"][:f["n_critic"]], [list(ac.critic.parameters())], lr=1.0)
    opt_a.load_state_dict(ck["optimizer_actor"])
    opt_c.load_state_dict(ck["optimizer_critic"])
    assert opt_a.param_groups[0]["lr"] == c["lr"] and int(opt_a.state_dev[0]) == int(fx["adam_step"])
    np.testing.assert_array_equal(opt_a.m[n_a:].numpy(), fx["adam_logstd_m"])
    np.testing.assert_array_equal(opt_a.v[n_a:].numpy(), fx["adam_logstd_v"])
    ours, ref = opt_a.state_dict(), ck["optimizer_actor"]
    assert len(ours["param_groups"]) == len(ref["param_groups"]) == 2
    assert [g["params"] for g in ours["param_groups"]] == [g["params"] for g in ref["param_groups"]]
    assert set(ours["state"].keys()) == set(ref["state"].keys())
    for k in ref["state"]:
        assert set(ref["state"][k].keys()) <= set(ours["state"][k].keys()) | {"step"}
        assert torch.equal(ours["state"][k]["exp_avg"], ref["state"][k]["exp_avg"])
        assert float(ours["state"][k]["step"]) == float(ref["state"][k]["step"])


def test_add_transitions_offline_reads_scene_step_npy(tmp_path):
    """storage.py:58-82: scene_*/step_*.npy dicts {tsdf, proprio_state, tea_obs} fill the DAgger ring row by row."""
    from partmanip_amd.algo_utils import RolloutStorage
    g = np.random.default_rng(0)
    rows = []
    for s in range(2):
        d = tmp_path / f"scene_{s:05d}"
        d.mkdir()
        for k in range(3):
            rec = dict(tsdf=g.standard_normal((2, 2, 2)).astype(np.float32),
                       proprio_state=g.standard_normal(3).astype(np.float32),
                       tea_obs=g.standard_normal(5).astype(np.float32))
            np.save(d / f"step_{k:05d}.npy", rec, allow_pickle=True)
            rows.append(rec)
    st = RolloutStorage(2, 2, 11, 4, "cpu", sampler="random", tea_obs_shape=5, max_length=10)    # ring of 4 rows
    st.add_transitions_offline(str(tmp_path), "cpu", add_proprio_obs=True)
    assert st.cur_buf_size == 4 and st.mix_buf_ind == 6 % 4 and st.last_episode_buf_ind == 2
    want = [np.concatenate([r["tsdf"].reshape(-1), r["proprio_state"]]) for r in rows]
    np.testing.assert_array_equal(st.observations[0].numpy(), want[4])        # rows 4,5 wrapped over rows 0,1
    np.testing.assert_array_equal(st.observations[1].numpy(), want[5])
    np.testing.assert_array_equal(st.observations[2].numpy(), want[2])
    np.testing.assert_array_equal(st.tea_obs[3].numpy(), rows[3]["tea_obs"])


def test_offline_preload_then_on_policy_rows_match_the_reference_ring(tmp_path):
    """Mixed BC + on-policy DAgger (storage.py:58-82 + :84-91, dagger.py:186-187): the ring contents and the three
    ring counters after the preload and after the on-policy steps equal what the REFERENCE's own storage produced
    on the same shards (tests/golden/make_golden.py gen_dagger_offline)."""
    from partmanip_amd.algo_utils import RolloutStorage
    from tests.golden import cases
    from tests.helpers import load_fixture, t
    c, fx = cases.DAGGER_OFFLINE_CASE, load_fixture("dagger_offline")
    cases.dagger_offline_write(c, str(tmp_path / "offline"))
    st = RolloutStorage(c["N"], c["buf_size"], c["D"] + c["proprio"], c["A"], "cpu", sampler=c["sampler"],
                        tea_obs_shape=c["O_t"], max_length=200)
    st.add_transitions_offline(str(tmp_path / "offline"), "cpu", add_proprio_obs=True)
    assert [st.mix_buf_ind, st.cur_buf_size, st.last_episode_buf_ind] == list(fx["off_state"])
    np.testing.assert_array_equal(st.observations.numpy(), fx["off_ring_obs"])
    np.testing.assert_array_equal(st.tea_obs.numpy(), fx["off_ring_tea"])
    on = cases.dagger_offline_online(c)
    for k in range(c["n_fill"]):
        st.add_transitions_dagger(t(on["stu"][k]), t(on["tea"][k]))
    assert [st.mix_buf_ind, st.cur_buf_size, st.last_episode_buf_ind] == list(fx["state"])
    np.testing.assert_array_equal(st.observations.numpy(), fx["ring_obs"])
    np.testing.assert_array_equal(st.tea_obs.numpy(), fx["ring_tea"])


def test_padded_cols_and_slab_count_host_side():
    """Host pieces of the small-step path: rows padded to 16 bytes carry their readable width for the weight-gradient
    launch, and the split-K slab count is lowered until every slab owns rows after rounding to the kernels' K-step of 32."""
    from types import SimpleNamespace
    from partmanip_amd import ops
    from partmanip_amd.algorithms.ppo import ppo
    v = ops.padded_cols(7, 53, torch.device("cpu"), zero=True)
    assert tuple(v.shape) == (7, 53) and v.stride(0) == 56 and v._pm_cols == 56 and float(v.abs().sum()) == 0.0
    w = ops.padded_cols(5, 1, torch.device("cpu"))
    assert w.stride(0) == 4 and w._pm_cols == 4 and ops.padded_cols(3, 64, torch.device("cpu")).stride(0) == 64
    for slabs in (1, 3, 8, 12, 64):
        me = SimpleNamespace(actor_critic=SimpleNamespace(GRAD_SLABS=slabs))
        for B in (8, 1023, 1024, 2048, 4100):
            S = ppo._slabs(me, B)
            chunk = (-(-B // S) + 31) // 32 * 32
            assert 1 <= S <= slabs and (S == 1 or (S - 1) * chunk < B) and (B >= 1024 or S == 1)
    assert ppo._slabs(SimpleNamespace(actor_critic=SimpleNamespace(GRAD_SLABS=3)), 2048) == 3


def test_pointnet2_without_a_set_abstraction_level_is_refused_at_construction():
    """`npoints: []` used to die with an IndexError inside the constructor; it is a configuration error and says so."""
    import pytest
    from partmanip_amd.algo_utils.network import PointNet2
    with pytest.raises(ValueError, match="at least one set-abstraction level"):
        PointNet2(3072, 10, dict(activation="tanh", npoints=[], radii=[], nsamples=[], mlps=[[64, 128]]), 0)
