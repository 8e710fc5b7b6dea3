"""End-to-end parity of the MI355X learner (partmanip_amd.algorithms.ppo / dagger, every step a
HIP kernel behind the C ABI) against (a) the golden vectors captured from the reference and
(b) the CPU oracle.  GPU box only.

Stated tolerances (fp32 everywhere; differences come only from summation order):
  loss traces / Train/* scalars ..... rtol 5e-4
  parameters after `update` ......... atol 5*lr*1e-2 (Adam turns a relative gradient error e into a
                                       parameter error ~ lr*e per step; near-zero-gradient elements
                                       at step 1 can move by up to lr -> compared on the 99.9 % quantile
                                       plus a hard bound of 2.5*lr*steps on the max)
  encoder features .................. 3e-6 relative to feature scale; argmax equal wherever the
                                       top-2 gap exceeds 1e-5
"""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.golden import cases
from tests.helpers import (load_fixture, t, ppo_cfg, ppo_rollout, flat_state, FakeEnv, FakeLogger, assert_update_matches, assert_close_rec,
                           record_margin)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# (rtol, atol) of the Train/* scalars against the reference's own values: 4x what the MI355X path was observed at (round 3,
# profiles/parity_margins.json; rounds 1-2 asserted rtol 5e-4 for all of them).  The surrogate of the golden cases is a mean
# of terms with |logp| ~ 1e2 in the exponent: one ulp there is 1e-5 relative in the ratio.
SCALAR_TOL = {"value_function_loss": (2e-6, 0.0), "surrogate_loss": (4e-4, 5e-6), "kl": (2e-5, 1e-7), "kl_max": (4e-5, 1e-7),
              "learning_rate": (1e-12, 0.0), "value_gt_return_mean": (2e-6, 2e-7), "value_gt_return_max": (1e-7, 0.0)}


def rel_err(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))


def make_ppo(c, name_o="O"):
    from partmanip_amd.algorithms import ppo
    env = FakeEnv(c["N"], {"normal_state": c[name_o]}, c["A"])
    with tempfile.TemporaryDirectory() as d:
        run = ppo(env, ppo_cfg(c, device=DEV), FakeLogger(d))
    sd = cases.actor_critic_state(c["net"], c[name_o], c["A"], c["action_std"], c["seed"])
    run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    return run


def fill_storage(run, c, fx):
    st = ppo_rollout(c, fx)
    for tt in range(c["T"]):
        run.storage.add_transitions(st["observations"][tt].to(DEV), st["actions"][tt].to(DEV),
                                    st["rewards"][tt, :, 0].to(DEV), st["dones"][tt, :, 0].to(DEV),
                                    st["succs"][tt, :, 0].to(DEV), st["values"][tt].to(DEV),
                                    st["actions_log_prob"][tt, :, 0].to(DEV), st["mu"][tt].to(DEV),
                                    st["sigma"][tt].to(DEV))
    return st


def check_params(fin, ref_flat, stride, lr, n_steps, init=None):
    """init: the initial state dict (name -> array, state_dict order) -> adds the per-tensor relative-L2 bound on the
    update (tests/helpers.py: a small tensor that moves the wrong way cannot hide in the whole-vector tail)."""
    if init is not None:
        return assert_update_matches(fin, ref_flat, init, lr, n_steps, stride)
    diff = np.abs(fin[::stride].astype(np.float64) - ref_flat.astype(np.float64))
    assert np.quantile(diff, 0.999) < 5e-2 * lr, (np.quantile(diff, 0.999), lr)
    assert diff.max() < 2.5 * lr * n_steps, (diff.max(), lr)


# ------------------------------------------------------------------------------- forward API
@pytest.mark.parametrize("name", list(cases.PPO_CASES))
def test_update_act_cri_matches_reference(name):
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    run = make_ppo(c)
    st = ppo_rollout(c, fx)
    logp, ent, val, mu, sig = run.actor_critic.update_act_cri(st["observations"].view(-1, c["O"]).to(DEV),
                                                              st["actions"].view(-1, c["A"]).to(DEV))
    assert_close_rec("mu", mu.cpu().numpy(), fx["fwd_mu"], rtol=2e-5, atol=3e-6)
    assert_close_rec("value", val.cpu().numpy(), fx["fwd_value"], rtol=2e-5, atol=3e-6)
    assert_close_rec("log_prob", logp.cpu().numpy(), fx["fwd_logp"], rtol=3e-5, atol=3e-4)
    assert_close_rec("entropy", ent.cpu().numpy(), fx["fwd_entropy"], rtol=1e-6)
    assert sig.shape == mu.shape


# ------------------------------------------------------------------------------- PPO update vs golden
@pytest.mark.parametrize("name", list(cases.PPO_CASES))
def test_ppo_update_matches_reference(name):
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    run = make_ppo(c)
    fill_storage(run, c, fx)
    run.storage.compute_returns(t(fx["last_values"]).to(DEV), c["gamma"], c["lam"])
    assert np.array_equal(run.storage.returns.cpu().numpy(), fx["returns"])
    if c["tricks"]["whole_adv_norm"]:
        np.testing.assert_allclose(run.storage.advantages.cpu().numpy(), fx["advantages"], rtol=2e-6, atol=5e-7)
    else:
        assert np.array_equal(run.storage.advantages.cpu().numpy(), fx["advantages"])
    if c["sampler"] == "random":
        torch.manual_seed(c["seed"])
    run.log_dict = {}
    run.update(c["it"])
    log = run.log_dict
    assert log["Train/kl_update_count"] == int(fx["log_kl_update_count"])
    for k in ("value_function_loss", "surrogate_loss", "kl", "kl_max", "learning_rate", "value_gt_return_mean",
              "value_gt_return_max"):
        assert_close_rec("Train/" + k, float(log["Train/" + k]), float(fx["log_" + k]), *SCALAR_TOL[k])
    fin = flat_state(run.actor_critic.state_dict())
    n_steps = len(fx["loss_trace"])
    check_params(fin, fx["final_flat"], int(fx["final_stride"]), c["lr"], n_steps,
                 init=cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"]))
    np.testing.assert_allclose([g["lr"] for g in run.optimizer_actor.param_groups], fx["lr_actor_groups"])
    np.testing.assert_allclose([g["lr"] for g in run.optimizer_critic.param_groups], fx["lr_critic_groups"])
    assert int(run.optimizer_actor.state_dev[0]) == int(fx["adam_step"])
    A = c["A"]
    np.testing.assert_allclose(run.optimizer_actor.m[-A:].cpu().numpy(), fx["adam_logstd_m"], rtol=2e-3, atol=1e-7)


def test_ppo_update_all_skipped_raises_like_reference():
    c = cases.case_copy(cases.PPO_CASES["ppo_mlp_default"])
    c["desired_kl"] = 1e-9
    fx = load_fixture("ppo_mlp_default")
    run = make_ppo(c)
    fill_storage(run, c, fx)
    run.storage.compute_returns(t(fx["last_values"]).to(DEV), c["gamma"], c["lam"])
    run.log_dict = {}
    with pytest.raises(ZeroDivisionError):
        run.update(1)
    # every actor mini-batch was skipped on the device: actor params and log_std untouched
    sd0 = cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"])
    sd1 = run.actor_critic.state_dict()
    for k in sd0:
        if k.startswith("actor.") or k == "log_std":
            assert np.array_equal(sd1[k].cpu().numpy(), sd0[k]), k
    assert int(run.optimizer_actor.state_dev[0]) == 0


def test_ppo_checkpoint_roundtrip():
    c, fx = cases.PPO_CASES["ppo_mlp_default"], load_fixture("ppo_mlp_default")
    from partmanip_amd.algorithms import ppo
    with tempfile.TemporaryDirectory() as d:
        run = make_ppo(c)
        run.save_ckpt_dir = d
        fill_storage(run, c, fx)
        run.storage.compute_returns(t(fx["last_values"]).to(DEV), c["gamma"], c["lam"])
        run.log_dict = {}
        run.update(1)
        run.save(3)
        ck = torch.load(f"{d}/model_3.pth", map_location="cpu", weights_only=False)
        assert set(ck) >= {"iteration", "model_state_dict", "optimizer_actor", "optimizer_critic", "total_steps",
                           "tricks", "obs_mode", "model_cfg"}
        assert list(ck["model_state_dict"].keys())[0] == "log_std"
        assert len(ck["optimizer_actor"]["param_groups"]) == 2 and len(ck["optimizer_critic"]["param_groups"]) == 1
        cfg = ppo_cfg(c, device=DEV)
        cfg["resume"] = f"{d}/model_3.pth"
        run2 = ppo(FakeEnv(c["N"], {"normal_state": c["O"]}, c["A"]), cfg, FakeLogger(d))
        assert run2.curr_iter == 3
        assert np.array_equal(flat_state(run2.actor_critic.state_dict()), flat_state(run.actor_critic.state_dict()))
        assert torch.equal(run2.optimizer_actor.m, run.optimizer_actor.m)
        assert int(run2.optimizer_actor.state_dev[0]) == int(run.optimizer_actor.state_dev[0])


# ------------------------------------------------------------------------------- PointNet encoder
@pytest.mark.parametrize("B,C,max_mean,sub_mean,proprio,P", [
    (6, 3, True, False, 0, 1024), (5, 3, False, False, 0, 1024), (4, 4, True, True, 0, 1024), (3, 3, True, True, 7, 1024),
    (2, 6, True, False, 0, 1024), (300, 3, True, False, 0, 1024),
    # generalised point_num (the reference hard-codes 1024, network.py:146)
    (3, 3, True, True, 5, 4096), (2, 4, False, False, 0, 2048), (3, 3, True, False, 0, 320)])
def test_pointnet_forward_backward(B, C, max_mean, sub_mean, proprio, P):
    from partmanip_amd.algo_utils import ActorCritic
    net = dict(name="PointNet", activation="tanh", max_mean=max_mean, sub_mean=sub_mean)
    if P != 1024:
        net["point_num"] = P
    O = P * C + proprio
    torch.manual_seed(B * 10 + C)
    ac = ActorCritic(O, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), proprio).to(DEV)
    f = ac.flat()
    g = torch.Generator().manual_seed(B)
    pts = torch.rand(B, P, C, generator=g) * 2 - 1 + (torch.rand(B, 1, C, generator=g) - 0.5)
    if B <= 8:
        pts[:, 100] = pts[:, 7]                               # duplicate points: max ties -> lowest index wins
    x = torch.cat([pts.reshape(B, -1), torch.randn(B, proprio, generator=g)], dim=1).contiguous()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ac.state_dict().items()}
    out_ref = R.pointnet_forward(p, "actor", net, x.clone(), proprio, point_num=P)
    dy = torch.randn(B, 10, generator=g)
    names = [k for k in p if k.startswith("actor.")]

    xd = x.to(DEV)
    out = ac.actor.hip_forward(xd)
    record_margin("encoder output (max abs / max|ref|)", rel_err(out, out_ref.detach()), 2e-5)
    assert rel_err(out, out_ref.detach()) < 2e-5
    # gradient reference with the pooling index pinned to the kernel's (checked against torch.max's below
    # wherever the top-2 gap is resolvable in fp32): see oracle.pointnet_forward's docstring
    out_pin = R.pointnet_forward(p, "actor", net, x.clone(), proprio, point_num=P,
                                 argmax_override=ac.actor._saved[2].cpu().long())
    assert rel_err(out_pin.detach(), out_ref.detach()) < 1e-6
    grads_ref = torch.autograd.grad((out_pin * dy).sum(), [p[k] for k in names])
    # pooled features + argmax against the oracle's own intermediate
    with torch.no_grad():
        pc = x[:, :P * C].reshape(B, P, C)
        if sub_mean:
            pc = torch.cat([pc[..., :3] - pc[..., :3].mean(dim=1, keepdim=True), pc[..., 3:]], dim=-1)
        h = torch.tanh(torch.nn.functional.linear(pc, p["actor.mlp.0.weight"], p["actor.mlp.0.bias"]))
        h = torch.tanh(torch.nn.functional.linear(h, p["actor.mlp.2.weight"], p["actor.mlp.2.bias"]))
        h = torch.nn.functional.linear(h, p["actor.mlp.4.weight"], p["actor.mlp.4.bias"])
        vmax, imax = h.max(dim=1)
        top2 = h.topk(2, dim=1)[0]
        gap = top2[:, 0] - top2[:, 1]
    _, feat, argmax = ac.actor._saved[:3]
    assert rel_err(feat[:, :512], vmax) < 3e-6
    if max_mean:
        assert rel_err(feat[:, 512:1024], h.mean(dim=1)) < 3e-6
    am = argmax.cpu().long()
    clear = gap > 1e-5
    assert torch.equal(am[clear], imax[clear])
    # how many (cloud, channel) pairs the equality above could NOT check (top-2 gap inside fp32 round-off), and how many of
    # those the kernel resolved differently from torch.max: a silent arg-max bug cannot hide in "near-tie" if this stays ~0
    n_sub, n_diff = int((~clear).sum()), int((am[~clear] != imax[~clear]).sum())
    print(f"arg-max: {n_sub} of {clear.numel()} (cloud, channel) pairs under the 1e-5 gap, {n_diff} of them resolved differently")
    record_margin("arg-max pairs under the 1e-5 top-2 gap / all pairs", n_sub / clear.numel(), 5e-2, differing=n_diff, pairs=int(clear.numel()))
    record_margin("arg-max pairs under the gap that differ from torch.max / all pairs", n_diff / clear.numel(), 2e-3)
    assert n_sub <= 5e-2 * clear.numel() and n_diff <= 2e-3 * clear.numel()
    if B <= 8:   # exact ties (duplicated point 7/100): the lower index must win, as torch.max does
        tie = (am == 100)
        assert not tie.any()

    ac.actor.hip_backward(dy.to(DEV))
    views, off = {}, 0
    for k, v in ac.actor.named_parameters():
        views["actor." + k] = f["grad_actor"][off:off + v.numel()].view(v.shape)
        off += v.numel()
    worst = 0.0
    for k, gr in zip(names, grads_ref):
        e = rel_err(views[k], gr)
        worst = max(worst, e)
        assert e < 1e-4, k
    record_margin("encoder parameter gradients (max abs / max|ref|)", worst, 1e-4)


@pytest.mark.parametrize("act", ["relu", "elu", "selu", "lrelu", "sigmoid", "crelu"])
@pytest.mark.parametrize("save_h2", [True, False])
def test_pointnet_other_activations_match_the_oracle(act, save_h2):
    """network.py:144,147-160 builds PointNet with ANY of get_activation's seven activations (network.py:7-24); the fused
    encoder kernels' generic instantiation (pm_act / pm_dact instead of the packed tanh) against the oracle: forward, pooled
    features, arg-max and every parameter gradient, through the saved-layer-2 backward (pn_bwd16_kernel) and the recomputing
    one (pn_bwd_kernel).  max + mean pooling, per-cloud centring, proprio columns."""
    from partmanip_amd.algo_utils import ActorCritic
    B, C, P, proprio = 5, 3, 1024, 6
    net = dict(name="PointNet", activation=act, max_mean=True, sub_mean=True, save_h2=save_h2)
    O = P * C + proprio
    torch.manual_seed(77)
    ac = ActorCritic(O, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), proprio).to(DEV)
    f = ac.flat()
    g = torch.Generator().manual_seed(5)
    pts = torch.rand(B, P, C, generator=g) * 2 - 1 + (torch.rand(B, 1, C, generator=g) - 0.5)
    x = torch.cat([pts.reshape(B, -1), torch.randn(B, proprio, generator=g)], dim=1).contiguous()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ac.state_dict().items()}
    out_ref = R.pointnet_forward(p, "actor", net, x.clone(), proprio, point_num=P)
    dy = torch.randn(B, 10, generator=g)
    names = [k for k in p if k.startswith("actor.")]
    out = ac.actor.hip_forward(x.to(DEV))
    record_margin(f"encoder output, activation {act} (max abs / max|ref|)", rel_err(out, out_ref.detach()), 2e-5)
    assert rel_err(out, out_ref.detach()) < 2e-5
    out_pin = R.pointnet_forward(p, "actor", net, x.clone(), proprio, point_num=P, argmax_override=ac.actor._saved[2].cpu().long())
    assert rel_err(out_pin.detach(), out_ref.detach()) < 1e-6          # the kernel's arg-max is (numerically) torch.max's
    grads_ref = torch.autograd.grad((out_pin * dy).sum(), [p[k] for k in names])
    ac.actor.hip_backward(dy.to(DEV))
    views, off = {}, 0
    for k, v in ac.actor.named_parameters():
        views["actor." + k] = f["grad_actor"][off:off + v.numel()].view(v.shape)
        off += v.numel()
    worst = 0.0
    for k, gr in zip(names, grads_ref):
        e = rel_err(views[k], gr)
        worst = max(worst, e)
        assert e < 1e-4, (act, k, e)
    record_margin(f"encoder parameter gradients, activation {act} (max abs / max|ref|)", worst, 1e-4)


def test_pointnet_full_batch_properties():
    """BASELINE size (B=2048 clouds x 1024 pts): permuting the points of every cloud leaves the max
    features bit-identical, the mean features equal to rounding, and maps argmax through the permutation."""
    from partmanip_amd.algo_utils import ActorCritic
    net = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False)
    torch.manual_seed(0)
    ac = ActorCritic(3072, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
    ac.flat()
    B = 2048
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.rand(B, 1024, 3, device=DEV, generator=g) * 2 - 1
    perm = torch.randperm(1024, device=DEV, generator=g)
    ac.actor.hip_forward(x.reshape(B, -1))
    _, f1, a1 = ac.actor._saved[:3]
    f1, a1 = f1.clone(), a1.clone()
    ac.actor.hip_forward(x[:, perm].reshape(B, -1).contiguous())
    _, f2, a2 = ac.actor._saved[:3]
    assert torch.equal(f1[:, :512], f2[:, :512])
    assert (f1[:, 512:] - f2[:, 512:]).abs().max() < 1e-5
    # equal unless two different points tie EXACTLY for a channel's max (then each ordering keeps its lowest index)
    assert float((perm[a2.long()] != a1.long()).float().mean()) < 2e-3
    assert torch.isfinite(f1).all()


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16x6"])
def test_pointnet_nan_point_poisons_its_cloud_like_torch(precision):
    """A NaN coordinate: torch.tanh propagates it through the shared MLP, `x.max(dim=1)` and `x.mean(dim=1)` return NaN for
    every channel of THAT cloud (network.py:175-181) -- the packed tanh of the fused encoders used to clamp it to -1 and the
    pooling's strict `>` skipped it.  The other clouds of the batch must not change by a bit."""
    from partmanip_amd.algo_utils import ActorCritic
    net = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False, precision=precision)
    torch.manual_seed(3)
    ac = ActorCritic(3072, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
    ac.flat()
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(6, 1024, 3, generator=g) * 2 - 1)
    clean = ac.actor.hip_forward(x.reshape(6, -1).to(DEV)).clone()
    feat_clean = ac.actor._saved[1].clone()
    x[2, 517, 1] = float("nan")
    x[4, 1023, 0] = float("nan")
    out = ac.actor.hip_forward(x.reshape(6, -1).to(DEV))
    feat = ac.actor._saved[1]
    p = {k: v.detach().cpu() for k, v in ac.state_dict().items()}
    ref = R.pointnet_forward(p, "actor", dict(net, precision="f32"), x.reshape(6, -1), 0)
    for b in range(6):
        if b in (2, 4):
            assert torch.isnan(feat[b]).all() and torch.isnan(out[b]).all() and torch.isnan(ref[b]).all(), b
        else:
            assert torch.equal(feat[b], feat_clean[b]) and torch.equal(out[b], clean[b]) and torch.isfinite(ref[b]).all(), b


def test_mlp_nan_observation_propagates_like_torch():
    from partmanip_amd.algo_utils import ActorCritic
    net = dict(name="MLP", hid_dim=[64, 64], activation="tanh")
    torch.manual_seed(4)
    ac = ActorCritic(20, 5, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
    ac.flat()
    x = torch.randn(300, 20)
    x[7, 3] = float("nan")
    out = ac.actor.hip_forward(x.to(DEV)).cpu()
    assert torch.isnan(out[7]).all() and torch.isfinite(out[torch.arange(300) != 7]).all()


# ------------------------------------------------------------------------------- DAgger vs golden
@pytest.mark.parametrize("name", list(cases.DAGGER_CASES))
def test_dagger_update_matches_reference(name, tmp_path, monkeypatch):
    from partmanip_amd.algorithms import ppo, dagger
    c, fx = cases.DAGGER_CASES[name], load_fixture(name)
    N, A = c["N"], c["A"]
    monkeypatch.chdir(tmp_path)
    np.save("teacher_reward.npy", np.linspace(0, 1, 200).astype(np.float32))
    tc = dict(net=c["tea_net"], N=N, T=1, n_updates=1, n_minibatches=1, tricks=dict(cases.TRICKS_DEFAULT),
              sampler="sequential", succ_value=None, lr=1e-3, desired_kl=0.1, lr_schedule="fixed", gamma=0.99,
              lam=0.95, epsilon_clip=0.2, action_std=0.5, max_iterations=10)
    tea_run = ppo(FakeEnv(N, {"normal_state": c["O_t"]}, A), ppo_cfg(tc, device=DEV), FakeLogger(str(tmp_path)))
    tea_run.actor_critic.load_state_dict(
        {k: t(v.copy()) for k, v in cases.actor_critic_state(c["tea_net"], c["O_t"], A, 0.5, c["seed"] + 1).items()})
    tea_run.save(1)
    env = FakeEnv(N, {"stu_mode": c["O_s"], "normal_state": c["O_t"], "proprio_state": c["proprio"]}, A)
    cfg = dict(num_envs=N, obs_mode="stu_mode",
               model=dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0, network=dict(c["stu_net"])),
               max_iterations=c["max_iterations"], n_steps=1, n_updates=c["n_updates"], n_minibatches=c["n_minibatches"],
               device=DEV, buf_size=c["buf_size"], reward_reset=True, add_proprio_obs=c["proprio"] > 0,
               offline_data_pth=None, eval_round=1, eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False,
               save_pose=False, save_video=False, lr_schedule=c["lr_schedule"], lr=c["lr"],
               teacher=str(tmp_path / "model_1.pth"), resume=None, pretrain=None, sampler=c["sampler"])
    run = dagger(env, cfg, FakeLogger(str(tmp_path)))
    run.student.load_state_dict({k: t(v.copy()) for k, v in
                                 cases.actor_critic_state(c["stu_net"], c["O_s"], A, c["action_std"], c["seed"], c["proprio"]).items()})
    raw = cases.dagger_raw_inputs(c)
    for k in range(c["n_fill"]):
        run.storage.add_transitions_dagger(t(raw["stu"][k]).to(DEV), t(raw["tea"][k]).to(DEV))
    assert (run.storage.mix_buf_ind, run.storage.cur_buf_size) == (int(fx["mix_buf_ind"]), int(fx["cur_buf_size"]))
    assert np.array_equal(run.storage.tea_obs.cpu().numpy(), fx["ring_tea"])
    np.testing.assert_allclose(run.teacher.act(run.storage.tea_obs).cpu().numpy(), fx["tea_act"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(run.student.act(run.storage.observations).cpu().numpy(), fx["stu_act0"], rtol=2e-5, atol=2e-6)
    torch.manual_seed(c["torch_seed"])
    run.log_dict = {}
    run.update(c["it"])
    np.testing.assert_allclose(run.log_dict["Train/dagger_loss"], float(fx["log_dagger_loss"]), rtol=1e-5)
    np.testing.assert_allclose(run.log_dict["Train/learning_rate"], float(fx["log_learning_rate"]), rtol=1e-12)
    fin = flat_state(run.student.state_dict())
    check_params(fin, fx["final_flat"], int(fx["final_stride"]), c["lr"], len(fx["loss_trace"]),
                 init=cases.actor_critic_state(c["stu_net"], c["O_s"], A, c["action_std"], c["seed"], c["proprio"]))
    run.save(2)
    ck = torch.load(str(tmp_path / "model_2.pth"), map_location="cpu", weights_only=False)
    assert set(ck) >= {"iteration", "model_state_dict", "optimizer_state_dict", "total_steps", "obs_mode", "teacher"}


def test_dagger_resume_and_load_pretrain_from_a_reference_checkpoint(tmp_path, monkeypatch):
    """A14 (dagger.py:98-120): `dagger(..., resume=<student checkpoint written by the REFERENCE's dagger.save>)` restores the student
    and the single Adam's state on the GPU; one more HIP update from there lands where the reference itself landed when IT resumed
    from that checkpoint (tests/golden/dagger_mlp_ckpt.npz); `load_pretrain` takes every tensor but log_std from the checkpoint."""
    import os
    from partmanip_amd.algorithms import ppo, dagger
    from tests.helpers import GOLDEN
    c, fx = cases.DAGGER_CASES["dagger_mlp"], load_fixture("dagger_mlp_ckpt")
    N, A = c["N"], c["A"]
    ckpt = os.path.join(GOLDEN, "ref_ckpt_dagger_mlp.pth")
    monkeypatch.chdir(tmp_path)
    np.save("teacher_reward.npy", np.linspace(0, 1, 200).astype(np.float32))
    tc = dict(net=c["tea_net"], N=N, T=1, n_updates=1, n_minibatches=1, tricks=dict(cases.TRICKS_DEFAULT), sampler="sequential",
              succ_value=None, lr=1e-3, desired_kl=0.1, lr_schedule="fixed", gamma=0.99, lam=0.95, epsilon_clip=0.2, action_std=0.5,
              max_iterations=10)
    tea_run = ppo(FakeEnv(N, {"normal_state": c["O_t"]}, A), ppo_cfg(tc, device=DEV), FakeLogger(str(tmp_path)))
    tea_run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in cases.actor_critic_state(c["tea_net"], c["O_t"], A, 0.5, c["seed"] + 1).items()})
    tea_run.save(1)
    env = FakeEnv(N, {"stu_mode": c["O_s"], "normal_state": c["O_t"], "proprio_state": c["proprio"]}, A)

    def make(resume=None, pretrain=None):
        cfg = dict(num_envs=N, obs_mode="stu_mode",
                   model=dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0, network=dict(c["stu_net"])),
                   max_iterations=c["max_iterations"], n_steps=1, n_updates=c["n_updates"], n_minibatches=c["n_minibatches"], device=DEV,
                   buf_size=c["buf_size"], reward_reset=True, add_proprio_obs=False, offline_data_pth=None, eval_round=1,
                   eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
                   lr_schedule=c["lr_schedule"], lr=c["lr"], teacher=str(tmp_path / "model_1.pth"), resume=resume, pretrain=pretrain,
                   sampler=c["sampler"])
        return dagger(env, cfg, FakeLogger(str(tmp_path)))

    run = make(resume=ckpt)
    assert run.curr_iter == c["it"] and run.total_envsteps == 4321
    np.testing.assert_array_equal(flat_state(run.student.state_dict()), fx["saved_flat"])
    raw = cases.dagger_raw_inputs(c)
    for k in range(c["n_fill"]):
        run.storage.add_transitions_dagger(t(raw["stu"][k]).to(DEV), t(raw["tea"][k]).to(DEV))
    torch.manual_seed(c["torch_seed"] + 1)
    run.log_dict = {}
    run.update(c["it"] + 1)
    np.testing.assert_allclose(run.log_dict["Train/dagger_loss"], float(fx["resume_log_dagger_loss"]), rtol=1e-5)
    np.testing.assert_allclose(run.log_dict["Train/learning_rate"], float(fx["resume_log_learning_rate"]), rtol=1e-12)
    ck = torch.load(ckpt, map_location="cpu", weights_only=False)
    check_params(flat_state(run.student.state_dict()), fx["resume_final_flat"], 1, c["lr"], len(fx["resume_loss_trace"]),
                 init={k: v.numpy() for k, v in ck["model_state_dict"].items()})
    # what we write back has the reference's layout: state for the actor's tensors only, the step counts the reference reached
    run.save(c["it"] + 1)
    ours = torch.load(str(tmp_path / f"model_{c['it'] + 1}.pth"), map_location="cpu", weights_only=False)
    assert sorted(ours["optimizer_state_dict"]["state"].keys()) == sorted(ck["optimizer_state_dict"]["state"].keys())
    assert [float(ours["optimizer_state_dict"]["state"][k]["step"]) for k in sorted(ours["optimizer_state_dict"]["state"])] == list(fx["resume_adam_steps"])
    # ---- load_pretrain (called by the constructor, dagger.py:78-79; and explicitly on a differently initialised student)
    run3 = make(pretrain=ckpt)
    got = flat_state(run3.student.state_dict())
    n_ls = run3.student.log_std.numel()
    np.testing.assert_array_equal(got[n_ls:], fx["saved_flat"][n_ls:])
    other = cases.actor_critic_state(c["stu_net"], c["O_s"], A, 0.3, c["seed"] + 50, c["proprio"])
    run3.student.load_state_dict({k: t(v.copy()) for k, v in other.items()})
    run3.load_pretrain(ckpt)
    np.testing.assert_array_equal(flat_state(run3.student.state_dict()), fx["pretrain_flat"])
    np.testing.assert_array_equal(run3.student.log_std.detach().cpu().numpy(), fx["pretrain_log_std"])


def test_bc_run_matches_reference(tmp_path):
    """`bc(...).run()` (dataset resident in HBM, HIP step) against the reference's own bc.run() on the same shards:
    same shuffled batches (same RNG consumption), per-iteration losses, lr schedule, final student."""
    from partmanip_amd.algorithms import bc
    c, fx = cases.BC_CASES["bc_mlp"], load_fixture("bc_mlp")
    cases.bc_write_dataset(c, str(tmp_path / "data"))
    trace = []

    class Log(FakeLogger):
        def info(self, log, it):
            trace.append((float(log["Train/bc_loss"]), float(log["Train/learning_rate"])))

    env = FakeEnv(4, {"tsdf": c["D"] + c["S"], "proprio_state": c["S"]}, c["A"])
    cfg = dict(num_envs=4, obs_mode="tsdf", model=dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0,
                                                        network=dict(c["net"])),
               max_iterations=c["max_iterations"], device=DEV, data_path=str(tmp_path / "data"),
               n_minibatches=c["n_minibatches"], add_proprio_obs=True, eval_round=1, eval_frequence=10 ** 9,
               save_frequence=2, test_only=False, save_pose=False, save_video=False, lr_schedule=c["lr_schedule"],
               lr=c["lr"], resume=None)
    run = bc(env, cfg, Log(str(tmp_path)))
    sd = cases.actor_critic_state(c["net"], c["D"] + c["S"], c["A"], c["action_std"], c["seed"])
    run.student.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    torch.manual_seed(c["torch_seed"])
    run.run()
    np.testing.assert_allclose([x[0] for x in trace], fx["loss_trace"], rtol=2e-6)
    np.testing.assert_allclose([x[1] for x in trace], fx["lr_trace"], rtol=1e-12)
    n_steps = c["max_iterations"] * 5
    check_params(flat_state(run.student.state_dict()), fx["final_flat"], 1, c["lr"], n_steps)
    ck = torch.load(str(tmp_path / "model_2.pth"), map_location="cpu", weights_only=False)
    assert set(ck) >= {"iteration", "model_state_dict", "optimizer_state_dict", "obs_mode", "total_steps", "tricks", "teacher"}
    cfg2 = dict(cfg, resume=str(tmp_path / "model_2.pth"))
    run2 = bc(env, cfg2, Log(str(tmp_path)))
    assert run2.curr_iter == 2


def test_dagger_small_buffer_is_noop(tmp_path, monkeypatch):
    from partmanip_amd.algo_utils import RolloutStorage
    st = RolloutStorage(4, 3, 8, 2, DEV, sampler="random", tea_obs_shape=5, max_length=10)
    st.add_transitions_dagger(torch.ones(4, 8, device=DEV), torch.ones(4, 5, device=DEV))
    assert st.cur_buf_size == 4 and st.mix_buf_ind == 4


def test_pointnet_saved_h2_and_recompute_backward_agree():
    """The training forward stores the layer-2 activations and the backward loads them (default); with
    `save_h2: False` the backward recomputes layer 2.  Same forward, gradients equal to fp32 round-off."""
    from partmanip_amd.algo_utils import ActorCritic
    grads = []
    x = (torch.rand(7, 2048 * 3, generator=torch.Generator().manual_seed(2)) * 2 - 1).to(DEV)
    dy = torch.randn(7, 10, generator=torch.Generator().manual_seed(3)).to(DEV)
    for save in (True, False):
        net = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=True, point_num=2048, save_h2=save)
        torch.manual_seed(8)
        ac = ActorCritic(2048 * 3, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), 0).to(DEV)
        f = ac.flat()
        out = ac.actor.hip_forward(x)
        assert (ac.actor._saved[3] is not None) == save
        ac.actor.hip_backward(dy)
        grads.append((out.clone(), f["grad_actor"][:f["n_actor"]].clone()))
        assert ac.actor(x).shape == out.shape and ac.actor._saved[3] is (ac.actor._saved[3])   # inference leaves _saved alone
    assert torch.equal(grads[0][0], grads[1][0])
    assert rel_err(grads[0][1], grads[1][1].double().cpu()) < 2e-6


def test_pointnet_backward_when_one_point_wins_every_channel():
    """Degenerate clouds: all points identical (every channel's arg-max is point 0 -> a 512-entry run inside one
    8-row block: the backward's >64-entry search path) and a cloud whose maxima all sit in its last point."""
    from partmanip_amd.algo_utils import ActorCritic
    net = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False)
    torch.manual_seed(21)
    ac = ActorCritic(3072, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), 0).to(DEV)
    f = ac.flat()
    g = torch.Generator().manual_seed(4)
    pts = torch.rand(3, 1024, 3, generator=g) * 0.02
    pts[0] = pts[0, :1]                                        # constant cloud
    pts[1, :1023] = pts[1, :1]                                 # constant except the last point ...
    pts[1, 1023] = torch.tensor([5.0, -4.0, 3.0])              # ... which saturates (most of) the maxima
    x = pts.reshape(3, -1).contiguous()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ac.state_dict().items()}
    out = ac.actor.hip_forward(x.to(DEV))
    am = ac.actor._saved[2].cpu().long()
    assert (am[0] == 0).all() and ((am[1] == 0) | (am[1] == 1023)).all() and (am[1] == 1023).sum() > 64
    out_pin = R.pointnet_forward(p, "actor", net, x.clone(), 0, argmax_override=am)
    assert rel_err(out, out_pin.detach()) < 2e-5
    dy = torch.randn(3, 10, generator=g)
    names = [k for k in p if k.startswith("actor.")]
    grads_ref = torch.autograd.grad((out_pin * dy).sum(), [p[k] for k in names])
    ac.actor.hip_backward(dy.to(DEV))
    off = 0
    for k, v in ac.actor.named_parameters():
        got = f["grad_actor"][off:off + v.numel()].view(v.shape)
        off += v.numel()
        assert rel_err(got, grads_ref[names.index("actor." + k)]) < 2e-4, k


# ------------------------------------------------------------------------------- Conv3D TSDF student
@pytest.mark.parametrize("name", ["conv3d_proprio", "conv3d_plain"])
def test_conv3dnet_forward_backward_matches_reference_module(name):
    """`network.name: Conv3DNet` on the HIP path (patch gather + MFMA Linear kernels + col2im) against the
    REFERENCE's own module (fixture from make_golden.gen_conv3d): outputs and parameter gradients."""
    from partmanip_amd.algo_utils import ActorCritic
    c, fx = cases.CONV3D_CASES[name], load_fixture(name)
    net = dict(name="Conv3DNet", activation="tanh")
    O = c["res"] ** 3 + c["proprio"]
    ac = ActorCritic(O, c["out"], dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), c["proprio"]).to(DEV)
    sd = cases.conv3d_state(c)
    ac.actor.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    f = ac.flat()
    inp = cases.conv3d_inputs(c)
    out = ac.actor.hip_forward(t(inp["x"]).to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), fx["out"], rtol=2e-5, atol=3e-6)
    ac.actor.hip_backward(t(inp["dy"]).to(DEV))
    off = 0
    for k, v in ac.actor.named_parameters():
        got = f["grad_actor"][off:off + v.numel()].cpu().numpy()[::7]
        off += v.numel()
        ref = fx["grad_" + k]
        assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-7, k


@pytest.mark.parametrize("act", ["relu", "lrelu", "elu", "selu", "sigmoid", "crelu"])
def test_conv3dnet_other_activations_match_the_oracle(act):
    """network.py:70 builds Conv3DNet with any of get_activation's seven (network.py:7-24): the layers' GEMM epilogues, the
    scatter data gradient and pm_col2im3d_f32 take the activation code (the input layer leaves its tanh stencil for the
    patch-matrix form).  Outputs and every parameter gradient against the restatement's autograd."""
    from partmanip_amd.algo_utils import ActorCritic
    c = cases.CONV3D_CASES["conv3d_proprio"]
    net = dict(name="Conv3DNet", activation=act)
    O = c["res"] ** 3 + c["proprio"]
    ac = ActorCritic(O, c["out"], dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), c["proprio"]).to(DEV)
    sd = cases.conv3d_state(c)
    ac.actor.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    f = ac.flat()
    inp = cases.conv3d_inputs(c)
    p = {"actor." + k: t(v.copy()).requires_grad_(True) for k, v in sd.items()}
    ref = R.net_forward(p, "actor", net, t(inp["x"]), c["proprio"])
    out = ac.actor.hip_forward(t(inp["x"]).to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=3e-6)
    names = list(p)
    grads = torch.autograd.grad((ref * t(inp["dy"])).sum(), [p[k] for k in names])
    ac.actor.hip_backward(t(inp["dy"]).to(DEV))
    off = 0
    for k, v in ac.actor.named_parameters():
        got = f["grad_actor"][off:off + v.numel()].view(v.shape)
        off += v.numel()
        assert rel_err(got, grads[names.index("actor." + k)]) < 2e-4, (act, k)


# ------------------------------------------------------------------------------- PointNet++ backbone
PN2_UNFUSED = dict(npoints=[128, 32], radii=[0.25, 0.5], nsamples=[16, 16], mlps=[[32, 32, 64], [64, 64, 128], [128, 256]])
# the shapes the fused SA kernels are instantiated for (pm_sa_fwd_f32 / pm_sa_bwd_f32); 130 / 33 centres make the
# last tile of both levels ragged
PN2_FUSED = dict(npoints=[130, 33], radii=[0.25, 0.5], nsamples=[32, 32], mlps=[[64, 64, 128], [128, 128, 256], [256, 512]])
# 64 centres at the last level: the group-all level runs fused too (csrc/sa_groupall.hip) -- the bench's shape
PN2_FUSED_GA = dict(npoints=[130, 64], radii=[0.25, 0.5], nsamples=[32, 32], mlps=[[64, 64, 128], [128, 128, 256], [256, 512]])


@pytest.mark.parametrize("B,C,proprio,shape", [(3, 3, 0, PN2_UNFUSED), (2, 5, 6, PN2_UNFUSED), (3, 3, 0, PN2_FUSED),
                                               (2, 5, 6, PN2_FUSED), (3, 3, 0, PN2_FUSED_GA), (2, 5, 6, PN2_FUSED_GA)])
def test_pointnet2_forward_backward(B, C, proprio, shape):
    """PointNet2 plug-in (FPS + ball query + grouping + shared MLP + max-pool; absent from the reference,
    parity unpinned): HIP path vs this build's CPU restatement -- sampled / grouped indices bit-exact,
    outputs to fp32 round-off, gradients with the pooling indices pinned."""
    from partmanip_amd.algo_utils import ActorCritic
    net = dict(name="PointNet2", activation="tanh", **shape)
    O = 1024 * C + proprio
    torch.manual_seed(11 * B + C)
    ac = ActorCritic(O, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), proprio).to(DEV)
    f = ac.flat()
    g = torch.Generator().manual_seed(B + C)
    pts = torch.rand(B, 1024, C, generator=g) * 2 - 1
    x = torch.cat([pts.reshape(B, -1), torch.randn(B, proprio, generator=g)], dim=1).contiguous()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ac.state_dict().items()}
    out_ref, aux, ref_args = R.pointnet2_forward(p, "actor", net, x.clone(), proprio, return_aux=True)
    out = ac.actor.hip_forward(x.to(DEV))
    saved = ac.actor._saved
    assert ac.actor._fused == [shape is not PN2_UNFUSED] * 2 and ac.actor._ga_fused == (shape is PN2_FUSED_GA)
    for l, (idx_c, idx_g) in enumerate(aux):
        assert torch.equal(saved[l][0].cpu().long(), idx_g), f"ball-query indices differ at level {l}"
    assert rel_err(out, out_ref.detach()) < 3e-5
    hip_args = [s[1].cpu().long() for s in saved]
    for a, b in zip(hip_args, ref_args):
        assert float((a != b).float().mean()) < 5e-3          # only near-ties may differ
    out_pin = R.pointnet2_forward(p, "actor", net, x.clone(), proprio, pool_args=hip_args)
    dy = torch.randn(B, 10, generator=g)
    names = [k for k in p if k.startswith("actor.")]
    grads_ref = torch.autograd.grad((out_pin * dy).sum(), [p[k] for k in names])
    ac.actor.hip_backward(dy.to(DEV))
    off = 0
    for k, v in ac.actor.named_parameters():
        got = f["grad_actor"][off:off + v.numel()].view(v.shape)
        off += v.numel()
        ref = grads_ref[names.index("actor." + k)]
        assert rel_err(got, ref) < 2e-4, k


def test_pointnet2_ppo_iteration_runs():
    from partmanip_amd.algorithms import ppo
    from partmanip_amd.feeder import FeederEnv, ScreenLogger
    net = dict(name="PointNet2", activation="tanh", npoints=[64, 16], radii=[0.4, 0.8], nsamples=[8, 8],
               mlps=[[16, 32], [32, 64], [64, 128]])
    cfg = dict(num_envs=8, obs_mode="depth_pc", succ_value=None,
               model=dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), max_iterations=2,
               n_steps=4, n_updates=2, n_minibatches=2, device=DEV, eval_round=1, eval_frequence=10 ** 9,
               save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False, lr_schedule="fixed", lr=1e-4,
               desired_kl=0.1, epsilon_clip=0.2, gamma=0.99, lam=0.95,
               tricks=dict(mini_adv_norm=False, whole_adv_norm=False, use_state_norm=False,
                           use_clipped_value_loss=False, use_grad_clip=True, max_grad_norm=0.5),
               sampler="sequential", resume=None)
    with tempfile.TemporaryDirectory() as d:
        env = FeederEnv(8, {"depth_pc": 3072}, 10, DEV, seed=2, max_episode_length=5)
        run = ppo(env, cfg, ScreenLogger(d, "g", "n", quiet=True))
        run.run()
    assert run.curr_iter == 2 and np.isfinite(float(run.log_dict["Train/surrogate_loss"]))


@pytest.mark.parametrize("shape", [PN2_UNFUSED, PN2_FUSED, PN2_FUSED_GA])
def test_pointnet2_neighbourhood_tables_reproduce_the_recomputed_forward(shape):
    """precompute_geometry() + use_geometry() (what ppo.update does once per rollout) must give the bit-identical
    forward as running FPS + ball query inside every forward, for slice and for index-tensor row selections."""
    from partmanip_amd.algo_utils import ActorCritic
    net = dict(name="PointNet2", activation="tanh", **shape)
    torch.manual_seed(5)
    ac = ActorCritic(3072, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), 0).to(DEV)
    ac.flat()
    obs = (torch.rand(12, 3072, device=DEV) * 2 - 1).contiguous()
    tabs = ac.actor.precompute_geometry(obs, chunk=5)
    ref = ac.actor.hip_forward(obs)
    ac.actor.use_geometry(tabs, (4, 6))
    assert torch.equal(ac.actor.hip_forward(obs[4:10]), ref[4:10])
    rows = torch.tensor([7, 0, 11, 3])
    ac.actor.use_geometry(tabs, rows)
    assert torch.equal(ac.actor.hip_forward(obs[rows.to(DEV)].contiguous()), ref[rows.to(DEV)])
    assert torch.equal(ac.actor.hip_forward(obs), ref)                       # tables are consumed by ONE forward


@pytest.mark.parametrize("sampler,shape", [("sequential", PN2_FUSED), ("random", PN2_FUSED), ("sequential", PN2_FUSED_GA)])
def test_ppo_update_pointnet2_follows_the_cpu_restatement(sampler, shape):
    """Whole `ppo.update` through the PointNet2 backbone (fused SA kernels + neighbourhood tables built once per
    rollout) against oracle/ref_cpu.py's ppo_update on the same rollout.  Parity unpinned (no PointNet2 in the
    reference): this checks the HIP path against this build's own restatement, same tolerances as the golden cases."""
    from partmanip_amd.algorithms import ppo
    c = cases.case_copy(cases.PPO_CASES["ppo_pn_maxmean"])
    fx = load_fixture("ppo_pn_maxmean")
    c["net"] = dict(name="PointNet2", activation="tanh", **shape)
    c["desired_kl"] = 10.0                 # the fixture's old policy is a PointNet: keep the KL early-stop out of the way
    c["sampler"] = sampler
    with tempfile.TemporaryDirectory() as d:
        torch.manual_seed(3)
        run = ppo(FakeEnv(c["N"], {"normal_state": c["O"]}, c["A"]), ppo_cfg(c, device=DEV), FakeLogger(d))
    assert run.actor_critic.actor._fused == [True, True] and run.actor_critic.actor._ga_direct == (shape is PN2_FUSED_GA)
    p = {k: v.detach().cpu().clone() for k, v in run.actor_critic.state_dict().items()}
    fill_storage(run, c, fx)
    run.storage.compute_returns(t(fx["last_values"]).to(DEV), c["gamma"], c["lam"])
    torch.manual_seed(c["seed"])
    run.log_dict = {}
    run.update(c["it"])
    st = ppo_rollout(c, fx)
    st["returns"], st["advantages"] = run.storage.returns.cpu(), run.storage.advantages.cpu()
    keys = ("observations", "actions", "values", "returns", "actions_log_prob", "advantages", "mu", "sigma")
    torch.manual_seed(c["seed"])
    out = R.ppo_update(p, {k: st[k] for k in keys}, ppo_cfg(c), c["it"])
    assert run.log_dict["Train/kl_update_count"] == out["log"]["Train/kl_update_count"]
    for k in ("value_function_loss", "surrogate_loss", "kl", "kl_max"):
        np.testing.assert_allclose(float(run.log_dict["Train/" + k]), float(out["log"]["Train/" + k]), rtol=1e-5,
                                   atol=5e-6, err_msg=k)
    check_params(flat_state(run.actor_critic.state_dict()), flat_state(p), 1, c["lr"], len(out["loss_trace"]))


def test_resume_from_reference_checkpoint_and_continue():
    """`ppo(... resume=<checkpoint written by the reference>)` restores model + both optimisers on the GPU, and one
    more HIP update from there follows the CPU oracle continuing from the same state."""
    import os
    from partmanip_amd.algorithms import ppo
    from tests.helpers import GOLDEN
    c, fx = cases.PPO_CASES["ppo_mlp_default"], load_fixture("ppo_mlp_default")
    c = cases.case_copy(c)
    c["desired_kl"] = 10.0        # the stale rollout is far from the resumed policy: keep the KL early-stop out of the way
    cfg = ppo_cfg(c, device=DEV)
    cfg["resume"] = os.path.join(GOLDEN, "ref_ckpt_ppo_mlp_default.pth")
    with tempfile.TemporaryDirectory() as d:
        run = ppo(FakeEnv(c["N"], {"normal_state": c["O"]}, c["A"]), cfg, FakeLogger(d))
    assert run.curr_iter == c["it"] and run.total_envsteps == 12345
    np.testing.assert_array_equal(flat_state(run.actor_critic.state_dict()), fx["final_flat"])
    # continue: same rollout once more on both sides
    fill_storage(run, c, fx)
    run.storage.compute_returns(t(fx["last_values"]).to(DEV), c["gamma"], c["lam"])
    run.log_dict = {}
    run.update(c["it"] + 1)
    ck = torch.load(cfg["resume"], map_location="cpu", weights_only=False)
    p = {k: v.clone() for k, v in ck["model_state_dict"].items()}
    ak, ckk = R.split_params(p)
    opt_a, opt_c = R.Adam([p[k] for k in ak] + [p["log_std"]], c["lr"]), R.Adam([p[k] for k in ckk], c["lr"])
    for opt, sd in ((opt_a, ck["optimizer_actor"]), (opt_c, ck["optimizer_critic"])):
        for i, st_ in sd["state"].items():
            opt.m[i], opt.v[i], opt.t[i] = st_["exp_avg"].clone(), st_["exp_avg_sq"].clone(), int(st_["step"])
    st = ppo_rollout(c, fx)
    st["returns"], st["advantages"] = t(fx["returns"]), t(fx["advantages"])
    keys = ("observations", "actions", "values", "returns", "actions_log_prob", "advantages", "mu", "sigma")
    out = R.ppo_update(p, {k: st[k] for k in keys}, ppo_cfg(c), c["it"] + 1, opt=(opt_a, opt_c))
    assert run.log_dict["Train/kl_update_count"] == out["log"]["Train/kl_update_count"]
    check_params(flat_state(run.actor_critic.state_dict()), flat_state(p), 1, c["lr"], len(out["loss_trace"]))


@pytest.mark.parametrize("B,C,max_mean,sub_mean", [(5, 3, True, False), (3, 4, False, True), (130, 3, True, False)])
def test_pointnet_bf16x3_forward(B, C, max_mean, sub_mean):
    """Opt-in split-bf16 encoder forward (`precision: bf16x3`): a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on bf16 MFMAs.
    Stated tolerance 1e-4 relative to the feature scale (fp32 path: 3e-6); arg-max equal where the top-2 gap > 1e-3."""
    from partmanip_amd.algo_utils import ActorCritic
    net = dict(name="PointNet", activation="tanh", max_mean=max_mean, sub_mean=sub_mean, precision="bf16x3")
    torch.manual_seed(B + C)
    ac = ActorCritic(1024 * C, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
    ac.flat()
    g = torch.Generator().manual_seed(B)
    x = (torch.rand(B, 1024, C, generator=g) * 2 - 1).reshape(B, -1).contiguous()
    p = {k: v.detach().cpu().clone() for k, v in ac.state_dict().items()}
    with torch.no_grad():
        out_ref = R.pointnet_forward(p, "actor", net, x.clone(), 0)
        pc = x.reshape(B, 1024, C)
        if sub_mean:
            pc = torch.cat([pc[..., :3] - pc[..., :3].mean(dim=1, keepdim=True), pc[..., 3:]], dim=-1)
        h = torch.tanh(torch.nn.functional.linear(pc, p["actor.mlp.0.weight"], p["actor.mlp.0.bias"]))
        h = torch.tanh(torch.nn.functional.linear(h, p["actor.mlp.2.weight"], p["actor.mlp.2.bias"]))
        h = torch.nn.functional.linear(h, p["actor.mlp.4.weight"], p["actor.mlp.4.bias"])
        vmax, imax = h.max(dim=1)
        top2 = h.topk(2, dim=1)[0]
    out = ac.actor.hip_forward(x.to(DEV))
    _, feat, argmax = ac.actor._saved[:3]
    assert rel_err(feat[:, :512], vmax) < 1e-4
    if max_mean:
        assert rel_err(feat[:, 512:1024], h.mean(dim=1)) < 1e-4
    assert rel_err(out, out_ref) < 2e-4
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert torch.equal(argmax.cpu().long()[clear], imax[clear])
    # the backward (fp32 kernels) runs on top of the bf16x3 forward
    ac.actor.hip_backward(torch.randn(B, 10, generator=g).to(DEV))
    assert torch.isfinite(ac.flat()["grad_actor"]).all()


@pytest.mark.parametrize("name", ["ppo_pn_maxmean", "ppo_pn_max"])
def test_ppo_update_bf16x3_forward_within_reference_tolerances(name):
    """The golden vision-PPO cases with the opt-in bf16x3 encoder forward: the SAME tolerances as the fp32 path
    on the Train/* scalars and on the parameters after the update (against the vectors captured from the reference)."""
    c = cases.case_copy(cases.PPO_CASES[name])
    c["net"] = dict(c["net"], precision="bf16x3")
    fx = load_fixture(name)
    run = make_ppo(c)
    fill_storage(run, c, fx)
    run.storage.compute_returns(t(fx["last_values"]).to(DEV), c["gamma"], c["lam"])
    run.log_dict = {}
    run.update(c["it"])
    log = run.log_dict
    assert log["Train/kl_update_count"] == int(fx["log_kl_update_count"])
    for k in ("value_function_loss", "surrogate_loss", "kl", "kl_max"):
        np.testing.assert_allclose(float(log["Train/" + k]), float(fx["log_" + k]), rtol=2e-4, atol=2e-6, err_msg=k)
    check_params(flat_state(run.actor_critic.state_dict()), fx["final_flat"], int(fx["final_stride"]), c["lr"],
                 len(fx["loss_trace"]))


@pytest.mark.parametrize("B,C,max_mean,sub_mean", [(5, 3, True, False), (3, 4, False, True), (130, 3, True, False)])
def test_pointnet_bf16x6_forward_has_fp32_class_error(B, C, max_mean, sub_mean):
    """Opt-in three-plane split-bf16 encoder forward (`precision: bf16x6`, six bf16 MFMAs per product block).  Claim:
    the error against an fp64 evaluation of the same network is that of the fp32 MFMA kernel.  Checked here: both
    kernels against fp64 on the same inputs -- the split kernel's mean error must not exceed 1.5x the fp32 kernel's,
    and it must meet the fp32 path's own tolerance (3e-6 of the feature scale); arg-max equal where the top-2 gap
    exceeds 1e-5 (the fp32 path's rule)."""
    from partmanip_amd.algo_utils import ActorCritic
    errs = {}
    for prec in ("f32", "bf16x6"):
        net = dict(name="PointNet", activation="tanh", max_mean=max_mean, sub_mean=sub_mean, precision=prec)
        torch.manual_seed(B + C)
        ac = ActorCritic(1024 * C, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
        ac.flat()
        g = torch.Generator().manual_seed(B)
        x = (torch.rand(B, 1024, C, generator=g) * 2 - 1).reshape(B, -1).contiguous()
        p = {k: v.detach().cpu().double() for k, v in ac.state_dict().items()}
        pc = x.double().reshape(B, 1024, C)
        if sub_mean:
            pc = torch.cat([pc[..., :3] - pc[..., :3].mean(dim=1, keepdim=True), pc[..., 3:]], dim=-1)
        h = torch.tanh(torch.nn.functional.linear(pc, p["actor.mlp.0.weight"], p["actor.mlp.0.bias"]))
        h = torch.tanh(torch.nn.functional.linear(h, p["actor.mlp.2.weight"], p["actor.mlp.2.bias"]))
        h = torch.nn.functional.linear(h, p["actor.mlp.4.weight"], p["actor.mlp.4.bias"])
        vmax, imax = h.max(dim=1)
        top2 = h.topk(2, dim=1)[0]
        ac.actor.hip_forward(x.to(DEV))
        _, feat, argmax = ac.actor._saved[:3]
        f = feat.cpu().double()
        scale = float(vmax.abs().mean())
        errs[prec] = (float((f[:, :512] - vmax).abs().mean()) / scale, float((f[:, :512] - vmax).abs().max()) / scale)
        if max_mean:
            assert float((f[:, 512:1024] - h.mean(dim=1)).abs().max()) / float(h.mean(dim=1).abs().mean()) < 1e-5
        clear = (top2[:, 0] - top2[:, 1]) > 1e-5
        assert torch.equal(argmax.cpu().long()[clear], imax[clear])
        if prec == "bf16x6":                                   # the fp32 backward runs on top of it
            ac.actor.hip_backward(torch.randn(B, 10, generator=g).to(DEV))
            assert torch.isfinite(ac.flat()["grad_actor"]).all()
    assert errs["bf16x6"][1] < 3e-6, errs
    assert errs["bf16x6"][0] <= 1.5 * errs["f32"][0] + 1e-9, errs


@pytest.mark.parametrize("B,C,max_mean,sub_mean,fwd", [(5, 3, True, False, "f32"), (3, 4, False, True, "f32"),
                                                       (300, 3, True, False, "f32"), (7, 6, True, True, "f32"),
                                                       (40, 3, True, False, "bf16x6")])
def test_pointnet_bf16x6_backward_has_fp32_class_error(B, C, max_mean, sub_mean, fwd):
    """Opt-in three-plane split-bf16 encoder BACKWARD (`precision_bwd: bf16x6`: dW2 = dz2^T h1 and dh1 = dz2 W2 on six bf16
    MFMAs per product block, csrc/pointnet_enc_bwd_bf6.h).  Claim: every parameter gradient has the fp32 MFMA kernel's error
    against an fp64 evaluation of the same network (pooling indices pinned to the forward's).  Checked: both backwards on the
    same forward (same saved layer 2 and arg-max) against fp64 autograd -- per tensor, the split kernel's relative error must
    not exceed 1.5 x the fp32 kernel's (+ 2e-7) or 2e-6 (the bias gradients are sums of 3e5 cancelling terms whose fp32
    error depends on the reduction tree: db1 at B = 300 comes out at 1.5e-6 here against 6e-7 -- every other tensor of every
    case sits at or below the fp32 kernel's error), and both meet the fp32 path's own 2e-4 gate; B = 300 walks two clouds per
    work-group, C = 6 the generic layer-1 width, the last case runs on top of the bf16x6 forward."""
    from partmanip_amd.algo_utils import ActorCritic
    g = torch.Generator().manual_seed(B + C)
    x = (torch.rand(B, 1024, C, generator=g) * 2 - 1).reshape(B, -1).contiguous()
    dy = torch.randn(B, 10, generator=g)
    grads, am = {}, None
    for bwd in ("f32", "bf16x6"):
        net = dict(name="PointNet", activation="tanh", max_mean=max_mean, sub_mean=sub_mean, precision=fwd, precision_bwd=bwd)
        torch.manual_seed(B + C)
        ac = ActorCritic(1024 * C, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
        f = ac.flat()
        ac.actor.hip_forward(x.to(DEV))
        a = ac.actor._saved[2].cpu().long()
        assert am is None or torch.equal(a, am)
        am = a
        ac.actor.hip_backward(dy.to(DEV))
        off, out = 0, {}
        for k, v in ac.actor.named_parameters():
            out[k] = f["grad_actor"][off:off + v.numel()].view(v.shape).double().cpu()
            off += v.numel()
        grads[bwd] = out
        p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in ac.state_dict().items()}
    net64 = dict(name="PointNet", activation="tanh", max_mean=max_mean, sub_mean=sub_mean)
    ref_out = R.pointnet_forward(p, "actor", net64, x.double(), 0, argmax_override=am)
    names = [k for k in p if k.startswith("actor.")]
    ref = dict(zip(names, torch.autograd.grad((ref_out * dy.double()).sum(), [p[k] for k in names])))
    for k in grads["f32"]:
        r = ref["actor." + k]
        e32 = float((grads["f32"][k] - r).norm() / r.norm())
        e6 = float((grads["bf16x6"][k] - r).norm() / r.norm())
        tol = max(1.5 * e32 + 2e-7, 2e-6)
        record_margin(f"bf16x6 backward vs fp64, relative L2 [{k}] (fp32 kernel: {e32:.2e})", e6, tol)
        assert e6 <= tol and e6 < 2e-4, (k, e6, e32)


@pytest.mark.parametrize("name", ["ppo_pn_maxmean", "ppo_pn_max"])
def test_ppo_update_bf16x6_forward_and_backward_within_reference_tolerances(name):
    """The golden vision-PPO cases with BOTH encoder directions on the three-plane split: the fp32 path's tolerances."""
    c = cases.case_copy(cases.PPO_CASES[name])
    c["net"] = dict(c["net"], precision="bf16x6", precision_bwd="bf16x6")
    fx = load_fixture(name)
    run = make_ppo(c)
    fill_storage(run, c, fx)
    run.storage.compute_returns(t(fx["last_values"]).to(DEV), c["gamma"], c["lam"])
    run.log_dict = {}
    run.update(c["it"])
    log = run.log_dict
    assert log["Train/kl_update_count"] == int(fx["log_kl_update_count"])
    for k in ("value_function_loss", "surrogate_loss", "kl", "kl_max"):
        np.testing.assert_allclose(float(log["Train/" + k]), float(fx["log_" + k]), rtol=5e-5, atol=5e-7, err_msg=k)
    check_params(flat_state(run.actor_critic.state_dict()), fx["final_flat"], int(fx["final_stride"]), c["lr"],
                 len(fx["loss_trace"]))


@pytest.mark.parametrize("name", ["ppo_pn_maxmean", "ppo_pn_max"])
def test_ppo_update_bf16x6_forward_within_reference_tolerances(name):
    """The golden vision-PPO cases with the bf16x6 encoder forward: the fp32 path's tolerances on the Train/* scalars and
    on the parameters after the update (against the vectors captured from the reference)."""
    c = cases.case_copy(cases.PPO_CASES[name])
    c["net"] = dict(c["net"], precision="bf16x6")
    fx = load_fixture(name)
    run = make_ppo(c)
    fill_storage(run, c, fx)
    run.storage.compute_returns(t(fx["last_values"]).to(DEV), c["gamma"], c["lam"])
    run.log_dict = {}
    run.update(c["it"])
    log = run.log_dict
    assert log["Train/kl_update_count"] == int(fx["log_kl_update_count"])
    for k in ("value_function_loss", "surrogate_loss", "kl", "kl_max"):
        np.testing.assert_allclose(float(log["Train/" + k]), float(fx["log_" + k]), rtol=5e-5, atol=5e-7, err_msg=k)
    check_params(flat_state(run.actor_critic.state_dict()), fx["final_flat"], int(fx["final_stride"]), c["lr"],
                 len(fx["loss_trace"]))


# ------------------------------------------------------------------------------- rollout side
def test_rollout_side_matches_reference():
    """`Normalization` (pm_rms_update_f32 / pm_rms_normalize_f32) and `ActorCritic.random_act_cri`
    (pm_gaussian_sample_f32) against the REFERENCE's own outputs (rollout_side.npz): running statistics to fp32
    round-off of the batch moments (fp64 sums here, torch's fp32 pairwise sums there), everything downstream to match."""
    from tests.test_oracle_golden import _rollout_inputs
    from partmanip_amd.algo_utils import Normalization, ActorCritic
    c, xs, obs = _rollout_inputs()
    fx = load_fixture("rollout_side")
    norm = Normalization(c["O"], DEV)
    for i, x in enumerate(xs):
        out = norm(t(x).to(DEV))
        st = np.stack([norm.running_ms.mean.cpu().numpy()[0], norm.running_ms.std.cpu().numpy()[0],
                       norm.running_ms.S.cpu().numpy()[0]])
        np.testing.assert_allclose(st, fx["norm_stats"][i], rtol=2e-6, atol=5e-7)
        np.testing.assert_allclose(out.cpu().numpy(), fx["norm_out"][i], rtol=5e-6, atol=2e-6)
    assert norm.running_ms.n == int(fx["n"]) and tuple(norm.running_ms.mean.shape) == (1, c["O"])
    np.testing.assert_allclose(norm(t(xs[2]).to(DEV), update=False).cpu().numpy(), fx["norm_frozen"], rtol=2e-6, atol=5e-7)
    assert norm.running_ms.n == int(fx["n"])                       # update=False leaves the statistics alone
    # save() / load() round trip keeps the kernels usable (RMS.py:20-34)
    norm2 = Normalization(c["O"], DEV)
    norm2.running_ms.load({k: (v.clone() if torch.is_tensor(v) else v) for k, v in norm.running_ms.save().items()})
    assert torch.equal(norm2(t(xs[1]).to(DEV), update=False), norm(t(xs[1]).to(DEV), update=False))

    ac = ActorCritic(c["O"], c["A"], dict(action_std=c["action_std"], action_activate="tanh", clipAction=c["max_action"],
                                          network=dict(c["net"]))).to(DEV)
    ac.load_state_dict({k: t(v.copy()) for k, v in
                        cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"]).items()})
    torch.manual_seed(c["torch_seed"])
    # the reference draws on the CPU generator; the same draw is injected so that every output is comparable
    real_normal = torch.normal
    try:
        torch.normal = lambda *a, **k: t(fx["eps"]).to(DEV)
        act, logp, val, mu, ls = ac.random_act_cri(t(obs).to(DEV))
        act_only = ac.random_act(t(obs).to(DEV))
    finally:
        torch.normal = real_normal
    np.testing.assert_allclose(mu.cpu().numpy(), fx["mu"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(val.cpu().numpy(), fx["value"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(act.cpu().numpy(), fx["actions"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(logp.cpu().numpy(), fx["logp"], rtol=2e-6, atol=1e-5)
    assert np.array_equal(ls.cpu().numpy(), fx["log_std_rows"])
    assert torch.equal(act_only, act)


@pytest.mark.parametrize("N,D", [(4096, 3072), (4096, 53), (1, 7), (333, 1000)])
def test_rms_kernels_against_restatement(N, D):
    """Observation-sized batches (4096 envs x 3072-d clouds / 53-d states), one row, ragged sizes: three updates with a
    drifting mean against the CPU restatement."""
    from partmanip_amd.algo_utils import Normalization
    g = torch.Generator().manual_seed(N + D)
    norm, ref = Normalization(D, DEV), R.RunningMeanStd(D)
    for i in range(3):
        x = torch.randn(N, D, generator=g) * (0.5 + i) + 3.0 * i + torch.linspace(-5, 5, D)
        out = norm(x.to(DEV))
        want = ref.normalize(x)
        np.testing.assert_allclose(norm.running_ms.mean.cpu().numpy(), ref.mean.numpy(), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(norm.running_ms.S.cpu().numpy(), ref.S.numpy(), rtol=5e-6, atol=2e-7)
        np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-5)


def test_pointnet2_forward_backward_is_bit_reproducible():
    """VERDICT r5 next #1 in the suite (the long form: tools/stress_pointnet2.py --reps 1000 --noise): the same PointNet++ forward +
    backward repeated under a noise stream -- output and every parameter gradient hashed -- must not change a bit (round 6: the
    level-1 dY sums run in a fixed order, no floating-point atomics)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "tools/stress_pointnet2.py", "--reps", "60", "--B", "256", "--noise"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "repetitions that differed: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
