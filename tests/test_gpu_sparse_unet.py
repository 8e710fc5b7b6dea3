"""`network.name: SparseUNet` on the HIP path (csrc/sparse_voxel.hip + the Linear kernels) against the restatement
oracle/ref_cpu.py::sparse_unet_forward (PARITY UNPINNED: the backbone is named by README.md:30 but absent from the reference
snapshot, README.md:23; the restatement itself is pinned to torch's dense conv3d U-Net on fully occupied grids,
tests/test_oracle_sparse_unet.py).  Integer tables bit-exact; features / gradients fp32 round-off."""
import tempfile

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.golden import cases
from tests.helpers import t, flat_state, FakeEnv, FakeLogger, ppo_cfg, assert_update_matches

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NET = dict(name="SparseUNet", activation="tanh", point_num=96, grid=14, channels=[16, 24, 32])


def _model(net, std=0.5):
    return dict(action_std=std, action_activate="tanh", clipAction=1.0, network=dict(net))


@pytest.mark.parametrize("dups", [False, True])
def test_geometry_tables_equal_the_restatement(dups):
    from partmanip_amd.algo_utils import ActorCritic
    P, Rg = NET["point_num"], NET["grid"]
    x = cases.sparse_clouds(5, P, Rg, 11, n_distinct=60 if dups else None, pad_tail=7 if dups else 0)
    ac = ActorCritic(4 * P, 3, _model(NET)).to(DEV)
    g = ac.actor.geometry(t(x).to(DEV))
    ref = R.sparse_unet_geometry(x, P, 4, Rg)
    assert g["rows"] == ref["rows"]
    np.testing.assert_array_equal(g["feat0"].cpu().numpy(), ref["feat0"])
    for k in ("nbr0", "nbr1", "nbr2"):
        np.testing.assert_array_equal(g[k].cpu().numpy(), ref[k], err_msg=k)
    for lv in ("l1", "l2"):
        for k in ("child", "parent", "parent_canon", "slot"):
            np.testing.assert_array_equal(g[lv][k].cpu().numpy().reshape(ref[lv][k].shape), ref[lv][k], err_msg=f"{lv}.{k}")
        np.testing.assert_array_equal(g[lv]["coords"][:, 1:].cpu().numpy(), np.concatenate(ref[lv]["coords"]))


NET_WIDE = dict(NET, channels=[32, 64, 64])        # every layer but conv0 has J*C % 32 == 0: all gathers fused into the GEMM loader


@pytest.mark.parametrize("proprio,dups,NET", [(0, False, NET), (5, True, NET), (0, True, NET_WIDE), (3, False, dict(NET_WIDE, fused_gather=False))])
def test_forward_and_parameter_gradients_match_the_restatement(proprio, dups, NET):
    from partmanip_amd.algo_utils import ActorCritic
    from partmanip_amd.autograd import backbone_apply
    P, Rg, A, B = NET["point_num"], NET["grid"], 6, 7
    O = 4 * P + proprio
    sd = cases.actor_critic_state(NET, O, A, 0.5, 41, proprio)
    ac = ActorCritic(O, A, _model(NET), proprio).to(DEV)
    ac.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    ac.flat()
    x = cases.sparse_clouds(B, P, Rg, 12, n_distinct=70 if dups else None, pad_tail=5 if dups else 0)
    if proprio:
        x = np.concatenate([x, np.random.default_rng(1).standard_normal((B, proprio)).astype(np.float32)], 1)
    w = np.random.default_rng(2).standard_normal((B, A)).astype(np.float32)
    xd = t(x).to(DEV)
    out = backbone_apply(ac.actor, xd)
    (out * t(w).to(DEV)).sum().backward()
    p = {k: torch.from_numpy(v.copy()).double().requires_grad_(True) for k, v in sd.items()}
    ref = R.sparse_unet_forward(p, "actor", NET, t(x).double(), proprio)
    (ref * t(w).double()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    for name, par in ac.actor.named_parameters():
        gr = p["actor." + name].grad
        err = float((par.grad.double().cpu() - gr).abs().max() / (gr.abs().max() + 1e-30))
        assert err < 2e-4, (name, err)
    # inference forward = training forward
    assert torch.equal(ac.actor(xd), out.detach())


def test_dagger_update_with_sparse_unet_student_matches_the_restatement(tmp_path, monkeypatch):
    """BASELINE cfg 5's shape in miniature: a SparseUNet student on 'depth_sparse' rows distils a frozen state MLP teacher."""
    from partmanip_amd.algorithms import ppo, dagger
    monkeypatch.chdir(tmp_path)
    P, Rg, A, N, O_t = NET["point_num"], NET["grid"], 6, 4, 20
    O_s = 4 * P
    tnet = dict(name="MLP", hid_dim=[32, 32], activation="tanh")
    tc = dict(net=tnet, N=N, T=1, n_updates=1, n_minibatches=1, tricks=dict(cases.TRICKS_DEFAULT), sampler="sequential",
              succ_value=None, lr=1e-3, desired_kl=0.1, lr_schedule="fixed", gamma=0.99, lam=0.95, epsilon_clip=0.2, action_std=0.5,
              max_iterations=10)
    tea = ppo(FakeEnv(N, {"normal_state": O_t}, A), ppo_cfg(tc, device=DEV), FakeLogger(str(tmp_path)))
    tsd = cases.actor_critic_state(tnet, O_t, A, 0.5, 52)
    tea.actor_critic.load_state_dict({k: t(v.copy()) for k, v in tsd.items()})
    tea.save(1)
    lr, n_fill = 1e-3, 5
    cfg = dict(num_envs=N, obs_mode="depth_sparse", model=_model(NET, 0.1), max_iterations=100, n_steps=1, n_updates=2,
               n_minibatches=2, device=DEV, buf_size=n_fill, reward_reset=False, add_proprio_obs=False, offline_data_pth=None,
               eval_round=1, eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
               lr_schedule="fixed", lr=lr, teacher=str(tmp_path / "model_1.pth"), resume=None, pretrain=None, sampler="sequential")
    env = FakeEnv(N, {"depth_sparse": O_s, "normal_state": O_t, "proprio_state": 0}, A)
    run = dagger(env, cfg, FakeLogger(str(tmp_path)))
    init = cases.actor_critic_state(NET, O_s, A, 0.1, 51)
    run.student.load_state_dict({k: t(v.copy()) for k, v in init.items()})
    obs = [cases.sparse_clouds(N, P, Rg, 60 + k, n_distinct=80, pad_tail=3) for k in range(n_fill)]
    tob = [np.random.default_rng(70 + k).standard_normal((N, O_t)).astype(np.float32) for k in range(n_fill)]
    for a, b in zip(obs, tob):
        run.storage.add_transitions_dagger(t(a).to(DEV), t(b).to(DEV))
    run.log_dict = {}
    run.update(1)
    stu = {k: t(v.copy()) for k, v in init.items()}
    ocfg = dict(model=_model(NET, 0.1), tea_model=_model(tnet), n_updates=2, n_minibatches=2, sampler="sequential", lr=lr,
                lr_schedule="fixed", max_iterations=100, proprio_shape=0)
    ref = R.dagger_update(stu, {k: t(v.copy()) for k, v in tsd.items()}, t(np.concatenate(obs)), t(np.concatenate(tob)), N * n_fill,
                          ocfg, 1)
    np.testing.assert_allclose(run.log_dict["Train/dagger_loss"], ref["log"]["Train/dagger_loss"], rtol=5e-4)
    assert_update_matches(flat_state(run.student.state_dict()), flat_state(stu), init, lr, len(ref["loss_trace"]))


def test_fused_and_materialised_gathers_agree_bit_for_bit():
    """`fused_gather`: the neighbour rows are gathered by the GEMM's LDS-DMA loader instead of being written to HBM first; the
    MFMA k-order is the same, so outputs and gradients are identical."""
    from partmanip_amd.algo_utils import ActorCritic
    from partmanip_amd.autograd import backbone_apply
    P, Rg, A, B = NET_WIDE["point_num"], NET_WIDE["grid"], 4, 9
    sd = cases.actor_critic_state(NET_WIDE, 4 * P, A, 0.5, 43)
    x = t(cases.sparse_clouds(B, P, Rg, 14, n_distinct=75, pad_tail=4)).to(DEV)
    w = torch.randn(B, A, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    res = []
    for fused in (True, False):
        ac = ActorCritic(4 * P, A, _model(dict(NET_WIDE, fused_gather=fused))).to(DEV)
        ac.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
        ac.flat()
        out = backbone_apply(ac.actor, x)
        (out * w).sum().backward()
        res.append((out.detach().clone(), [p.grad.clone() for p in ac.actor.parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
