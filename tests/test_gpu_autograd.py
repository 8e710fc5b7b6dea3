"""INTEGRATION.md "Level 1.5": a training loop shaped like the REFERENCE's own `ppo.update` (ppo.py:307-411) / `dagger.update`
(dagger.py:299-337) -- written fresh here -- that differentiates through `ActorCritic.update_act_cri` / `update_act` with
`loss.backward()`, clips with `torch.nn.utils.clip_grad_norm_` and steps `torch.optim.Adam`, drives the HIP backbones through
the torch.autograd bridge (partmanip_amd/autograd.py) to the same golden results as the fused runners."""
import numpy as np
import pytest
import torch

from tests.golden import cases
from tests.helpers import load_fixture, t, ppo_rollout, ppo_model_cfg, flat_state, assert_update_matches

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference_shaped_ppo_update(ac, st, c):
    """Two optimisers as ppo.py:73-74; actor epochs then critic epochs over the sequential mini-batches."""
    opt_a = torch.optim.Adam([{"params": ac.actor.parameters()}, {"params": [ac.log_std]}], lr=c["lr"])
    opt_c = torch.optim.Adam(ac.critic.parameters(), lr=c["lr"])
    flat = {k: v.reshape(-1, v.shape[-1]).to(DEV) for k, v in st.items()}
    n = flat["observations"].shape[0]
    mb = min(n // c["n_minibatches"], 2048)
    batches = [torch.arange(k * mb, (k + 1) * mb, device=DEV) for k in range(n // mb)]
    tr = c["tricks"]
    stats = dict(surr=0.0, kl=0.0, count=0, v=0.0, nv=0)
    for _ in range(c["n_updates"]):
        for idx in batches:
            logp, _, _, mu, sig = ac.update_act_cri(flat["observations"][idx], flat["actions"][idx])
            adv = flat["advantages"][idx]
            if tr["mini_adv_norm"]:
                adv = (adv - adv.mean()) / (adv.std() + 1e-8)
            old_mu, old_sig = flat["mu"][idx], flat["sigma"][idx]
            with torch.no_grad():
                kl = (sig - old_sig + (old_sig.exp().square() + (old_mu - mu).square()) / (2.0 * sig.exp().square()) - 0.5).sum(-1).mean()
            if float(kl) > c["desired_kl"]:
                continue
            ratio = torch.exp(logp - flat["actions_log_prob"][idx].squeeze(-1))
            a1 = adv.squeeze(-1)
            loss = torch.max(-a1 * ratio, -a1 * ratio.clamp(1.0 - c["epsilon_clip"], 1.0 + c["epsilon_clip"])).mean()
            opt_a.zero_grad()
            loss.backward()
            if tr["use_grad_clip"]:
                torch.nn.utils.clip_grad_norm_(ac.actor.parameters(), tr["max_grad_norm"])
            opt_a.step()
            stats["surr"] += float(loss)
            stats["kl"] += float(kl)
            stats["count"] += 1
    for _ in range(c["n_updates"]):
        for idx in batches:
            _, _, value, _, _ = ac.update_act_cri(flat["observations"][idx], flat["actions"][idx])
            ret, old_v = flat["returns"][idx], flat["values"][idx]
            if tr["use_clipped_value_loss"]:
                d = (c["epsilon_clip"] * old_v).abs().mean()
                loss = (value - (old_v + (ret - old_v).clamp(-d, d))).pow(2).mean()
            else:
                loss = (ret - value).pow(2).mean()
            opt_c.zero_grad()
            loss.backward()
            if tr["use_grad_clip"]:
                torch.nn.utils.clip_grad_norm_(ac.critic.parameters(), tr["max_grad_norm"])
            opt_c.step()
            stats["v"] += float(loss)
            stats["nv"] += 1
    return stats


@pytest.mark.parametrize("name", ["ppo_mlp_default", "ppo_mlp_allon", "ppo_mlp_klskip", "ppo_pn_maxmean"])
def test_reference_shaped_update_through_autograd_bridge_matches_golden(name):
    from partmanip_amd.algo_utils import ActorCritic
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    if c["sampler"] != "sequential":
        pytest.skip("sequential cases only")
    sd = cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"])
    ac = ActorCritic(c["O"], c["A"], ppo_model_cfg(c)).to(DEV)
    ac.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    ac.autograd = True
    st = ppo_rollout(c, fx)
    st = {k: st[k] for k in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma")}
    st["returns"], st["advantages"] = t(fx["returns"]), t(fx["advantages"])
    stats = _reference_shaped_ppo_update(ac, st, c)
    assert stats["count"] == int(fx["log_kl_update_count"])
    np.testing.assert_allclose(stats["surr"] / stats["count"], float(fx["log_surrogate_loss"]), rtol=5e-4, atol=5e-6)
    np.testing.assert_allclose(stats["kl"] / stats["count"], float(fx["log_kl"]), rtol=1e-5, atol=2e-7)
    np.testing.assert_allclose(stats["v"] / stats["nv"], float(fx["log_value_function_loss"]), rtol=1e-5)
    assert_update_matches(flat_state(ac.state_dict()), fx["final_flat"], sd, c["lr"], len(fx["loss_trace"]), int(fx["final_stride"]))


def test_autograd_bridge_gradients_equal_the_fused_backward():
    """d(sum(logp * g) + sum(value * h)) / d(parameters) through autograd == what the fused kernels produce for the same
    upstream gradients; and a second training forward invalidates an older graph loudly."""
    from partmanip_amd.algo_utils import ActorCritic
    c = cases.PPO_CASES["ppo_mlp_default"]
    sd = cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"])
    ac = ActorCritic(c["O"], c["A"], ppo_model_cfg(c)).to(DEV)
    ac.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    ac.autograd = True
    g = torch.Generator(device=DEV).manual_seed(0)
    obs = torch.randn(64, c["O"], device=DEV, generator=g)
    act = (torch.rand(64, c["A"], device=DEV, generator=g) * 1.8 - 0.9)
    logp, ent, value, mu, sig = ac.update_act_cri(obs, act)
    assert logp.requires_grad and value.requires_grad and mu.requires_grad and sig.requires_grad
    wl, wv = torch.randn(64, device=DEV, generator=g), torch.randn(64, 1, device=DEV, generator=g)
    ((logp * wl).sum() + 0.3 * ent.sum() + (value * wv).sum()).backward()
    # the same derivative with plain torch ops on the same parameters (double precision reference)
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in ac.state_dict().items()}
    def mlp(prefix, x):
        for i in (0, 2, 4):
            x = torch.tanh(x @ p[f"{prefix}.model.{i}.weight"].t() + p[f"{prefix}.model.{i}.bias"])
        return x @ p[f"{prefix}.model.6.weight"].t() + p[f"{prefix}.model.6.bias"]
    x64 = obs.double().cpu()
    m = mlp("actor", x64)
    s2 = p["log_std"].exp() ** 2
    xr = torch.atanh((act.double().cpu()).clamp(-1 + 1e-5, 1 - 1e-5))
    lp = (-0.5 * ((xr - m) / s2) ** 2 - s2.log() - 0.5 * np.log(2 * np.pi)).sum(-1)
    en = (0.5 * (1 + np.log(2 * np.pi)) + s2.log()).sum() * torch.ones(64, dtype=torch.float64)
    ((lp * wl.double().cpu()).sum() + 0.3 * en.sum() + (mlp("critic", x64) * wv.double().cpu()).sum()).backward()
    for (k, q), par in zip(p.items(), ac.state_dict(keep_vars=True).values()):
        assert par.grad is not None, k
        ref = q.grad
        err = float((par.grad.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-12))
        assert err < 2e-4, (k, err)
    l1, _, _, _, _ = ac.update_act_cri(obs, act)
    ac.update_act_cri(obs, act)                                           # a newer training forward of the same backbones
    with pytest.raises(RuntimeError, match="another training forward"):
        l1.sum().backward()


def test_update_act_autograd_matches_dagger_fixture():
    """dagger.py:312-319 shaped step: student.update_act -> mse to the teacher action -> backward -> Adam over ALL student params."""
    from partmanip_amd.algo_utils import ActorCritic
    name = "dagger_pn"
    c, fx = cases.DAGGER_CASES[name], load_fixture(name)
    A = c["A"]
    model = lambda net, std: dict(action_std=std, action_activate="tanh", clipAction=1.0, network=dict(net))
    stu = ActorCritic(c["O_s"], A, model(c["stu_net"], c["action_std"]), c["proprio"]).to(DEV)
    init = cases.actor_critic_state(c["stu_net"], c["O_s"], A, c["action_std"], c["seed"], c["proprio"])
    stu.load_state_dict({k: t(v.copy()) for k, v in init.items()})
    stu.autograd = True
    tea = ActorCritic(c["O_t"], A, model(c["tea_net"], 0.5)).to(DEV)
    tea.load_state_dict({k: t(v.copy()) for k, v in cases.actor_critic_state(c["tea_net"], c["O_t"], A, 0.5, c["seed"] + 1).items()})
    raw = cases.dagger_raw_inputs(c)
    ring_obs = torch.cat([t(x) for x in raw["stu"]]).to(DEV)
    ring_tea = torch.cat([t(x) for x in raw["tea"]]).to(DEV)
    opt = torch.optim.Adam(stu.parameters(), lr=c["lr"])
    losses = []
    for ep in range(c["n_updates"]):
        for idx in fx["index_lists"][ep]:
            idx = torch.as_tensor(idx, device=DEV)
            with torch.no_grad():
                tea_act = tea.act(ring_tea[idx])
            loss = (tea_act - stu.update_act(ring_obs[idx].contiguous())).pow(2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
    np.testing.assert_allclose(losses, fx["loss_trace"], rtol=1e-5, atol=1e-9)
    assert_update_matches(flat_state(stu.state_dict()), fx["final_flat"], init, c["lr"], len(losses), int(fx["final_stride"]))


@pytest.mark.parametrize("act", ["relu", "lrelu", "elu", "selu", "sigmoid", "crelu", "tanh"])
def test_mlp_activation_set_forward_and_gradients(act):
    """network.py:7-24: every activation `get_activation` knows runs as an epilogue of the Linear kernels (forward) and of the
    data-gradient kernels (derivative through the activation OUTPUT); values and parameter gradients against torch in fp64."""
    import torch.nn.functional as F
    from partmanip_amd.algo_utils import ActorCritic
    fn = dict(relu=F.relu, crelu=F.relu, lrelu=lambda v: F.leaky_relu(v, 0.01), elu=F.elu, selu=F.selu, sigmoid=torch.sigmoid,
              tanh=torch.tanh)[act]
    O, A, B = 24, 6, 96
    net = dict(name="MLP", hid_dim=[32, 48], activation=act)
    torch.manual_seed(11)
    ac = ActorCritic(O, A, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
    ac.autograd = True
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, O, device=DEV, generator=g)
    w = torch.randn(B, A, device=DEV, generator=g)
    ac.flat()
    from partmanip_amd.autograd import backbone_apply
    mu = backbone_apply(ac.actor, x)
    (mu * w).sum().backward()
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in ac.actor.state_dict().items()}
    h = x.double().cpu()
    for i in (0, 2):
        h = fn(h @ p[f"model.{i}.weight"].t() + p[f"model.{i}.bias"])
    ref = h @ p["model.4.weight"].t() + p["model.4.bias"]
    (ref * w.double().cpu()).sum().backward()
    assert float((mu.detach().double().cpu() - ref.detach()).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    for (k, q), par in zip(p.items(), ac.actor.parameters()):
        err = float((par.grad.double().cpu() - q.grad).abs().max() / (q.grad.abs().max() + 1e-12))
        assert err < 3e-4, (act, k, err)


def test_plug_in_encoders_reject_other_activations_loudly():
    """The reference's own backbones (MLP, PointNet) take every activation of network.py:7-24; the plug-in backbones that
    are absent from the reference (PointNet2, SparseUNet), the Conv3DNet stencils and the split-bf16 PointNet forwards are
    tanh kernels and say so instead of computing something else."""
    from partmanip_amd.algo_utils import ActorCritic
    model = lambda net: dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)
    for net, O in ((dict(name="PointNet2", activation="relu"), 3072),
                   (dict(name="SparseUNet", activation="elu", point_num=96, grid=14), 4 * 96),
                   (dict(name="PointNet", activation="relu", max_mean=True, sub_mean=False, precision="bf16x6"), 3072)):
        with pytest.raises(NotImplementedError, match="tanh"):
            ActorCritic(O, 4, model(net))
    ActorCritic(3072, 4, model(dict(name="PointNet", activation="relu", max_mean=True, sub_mean=False)))      # fp32 kernels: fine
    # the split-bf16 backward is a tanh kernel on the forward's saved layer 2: both conditions are checked at construction
    with pytest.raises(NotImplementedError, match="tanh"):
        ActorCritic(3072, 4, model(dict(name="PointNet", activation="elu", max_mean=True, sub_mean=False, precision_bwd="bf16x6")))
    with pytest.raises(ValueError, match="save_h2"):
        ActorCritic(3072, 4, model(dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False, precision_bwd="bf16x6",
                                        save_h2=False)))
