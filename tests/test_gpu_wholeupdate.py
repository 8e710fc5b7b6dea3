"""Whole-update parity at the BENCH shapes (VERDICT r4 "next" #1): the three numbers `bench.py` quotes for the point-cloud
backbones were checked in pieces (one B = 2048 step, N = 4 fixtures, kernel-vs-kernel at 2048 clouds).  Here each whole update
runs on the HIP path and on the oracle -- `oracle/ref_cpu.py::ppo_update` / `dagger_update`, plain torch, EXECUTED ON THE MI355X
THROUGH ATen (the restatement is device-agnostic; on the host cores these updates take hours) -- in fp32 and in fp64:

  (a) BASELINE cfg 3, one whole iteration: 4096 envs x 8 steps x 1024 points, PointNet, 5 epochs x 16 mini-batches of 2048,
      actor loop then critic loop (/root/reference/algorithms/ppo.py:314-384): 80 + 80 optimiser steps;
  (b) the same rollout through `network.name: PointNet2` (bench `secondary.vision_pn2`, BASELINE configs[2] as worded);
      the FPS centres and ball-query tables of all 32 768 clouds must equal the restatement's bit for bit;
  (c) one `dagger.update` of the SparseUNet student at 2048-cloud mini-batches of 4096-voxel clouds (bench `--workload dagger
      --student sparse_unet`; /root/reference/algorithms/dagger.py:299-337), random sampler.

Bracket (as tests/test_gpu_fullsize.py::test_cfg2_full_ppo_update_matches_oracle): after tens of Adam steps two correct fp32
evaluations differ by round-off amplified through sign(g)-like first steps and max-pool winners of near-tied channels, so the
fp32 oracle is itself some distance from its own fp64 evaluation; the HIP path must be as close to the fp64 trajectory as the fp32
oracle is (per tensor, x BRACKET), returns and integer tables bit-exact, the same steps taken, loss scalars tight.  Observed
margins go to gpurun_out/parity_margins.jsonl -> profiles/parity_margins.json."""
import gc
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.golden import cases
from tests.helpers import t, flat_state, FakeEnv, FakeLogger, per_tensor_update_error, assert_close_rec, record_margin

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# Round 6: the HIP side has no floating-point atomics any more (the level-1 dY sums run in a fixed order over the plan's inverse
# table, csrc/sa_fused.hip), so its trajectory is bit-reproducible; what still varies run to run is the ORACLE's ATen index_add.
# 2 x for all three workloads again (round 5 needed 3 x for the PointNet++ iteration).  One fp32 evaluation is ONE draw of a noisy
# yardstick -- the fp32 oracle's own distance to fp64 on `critic.sa.2.0.bias` ranged over 5.4e-4 .. 1.09e-3 in six runs -- and the
# test takes the MAX over ~60 tensors, so the PointNet++ iteration measures the yardstick three times: the oracle with 256-, 128- and
# 192-cloud pieces (same update, other summation orders), per tensor the worst of the three (`second_oracle_chunk`).
# Observed (profiles/round6_pn2_bracket2_runs.txt): worst tensor at 0.61 .. 0.81 of the bound before the consumer kernel changed
# the summation order of dfeat / dW1f, one run at 1.03 with a single-draw yardstick after it -- which is what the second draw is for.
# A wrong gradient scores 10 x and more.
BRACKET = 2.0

TRICKS = dict(mini_adv_norm=False, whole_adv_norm=False, use_state_norm=False, use_clipped_value_loss=False,
              use_grad_clip=True, max_grad_norm=0.5)


def _ppo_cfg(net, N, T, device, lr=5e-5):
    """bench.py::make_cfg: ppo.yaml's hyper-parameters (5 epochs, n_minibatches 8 -> the 2048-row cap, storage.py:127)."""
    return dict(num_envs=N, obs_mode="normal_state", succ_value=None,
                model=dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=dict(net)),
                max_iterations=200000, n_steps=T, n_updates=5, n_minibatches=8, device=device, eval_round=1,
                eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
                lr_schedule="fixed", lr=lr, desired_kl=0.1, epsilon_clip=0.2, gamma=0.99, lam=0.95, tricks=dict(TRICKS),
                sampler="sequential", resume=None)


def _exact_fp32():
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    torch.set_float32_matmul_precision("highest")


def _free():
    gc.collect()
    torch.cuda.empty_cache()


def _clouds(N, T, seed):
    """1024 points ~ U([-1,1]^3) + a per-env translation U(-0.5,0.5), (T, N, 3072); seeded device generator."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    pts = torch.rand(T, N, 1024, 3, device=DEV, generator=g) * 2 - 1
    pts = pts + (torch.rand(T, N, 1, 3, device=DEV, generator=g) - 0.5)
    return pts.reshape(T, N, 3072).contiguous()


def _rollout(p, model_cfg, obs, seed, geom=None, chunk=2048):
    """A rollout sampled from the CURRENT policy by the oracle on the device (ratio ~ 1, KL ~ 0 at step 0, as in the reference)."""
    T, N, O = obs.shape
    A = p["log_std"].numel()
    g = torch.Generator(device=DEV).manual_seed(seed)
    flat = obs.reshape(T * N, O)
    eps = torch.randn(T * N, A, device=DEV, generator=g)
    outs = []
    with torch.no_grad():
        for lo in range(0, T * N, chunk):
            if geom is None:
                outs.append(R.random_act_cri(p, model_cfg, flat[lo:lo + chunk], eps[lo:lo + chunk]))
            else:                                       # PointNet2: this chunk's rows of the rollout's tables
                mu = R.net_forward(p, "actor", model_cfg["network"], flat[lo:lo + chunk], 0, [(c[lo:lo + chunk], q[lo:lo + chunk]) for c, q in geom])
                sig2 = p["log_std"].exp() * p["log_std"].exp()
                x = mu + sig2 * eps[lo:lo + chunk]
                logp, _ = R.gaussian_logp_entropy(mu, p["log_std"], x)
                val = R.net_forward(p, "critic", model_cfg["network"], flat[lo:lo + chunk], 0, [(c[lo:lo + chunk], q[lo:lo + chunk]) for c, q in geom])
                outs.append((R.action_activation(x, "tanh", 1.0), logp, val, mu, p["log_std"].repeat(mu.shape[0], 1)))
    act, logp, val, mu, ls = (torch.cat([o[i] for o in outs]) for i in range(5))
    rew = torch.randn(T, N, 1, device=DEV, generator=g)
    dones = torch.rand(T, N, 1, device=DEV, generator=g) < 0.02
    succs = dones & (torch.rand(T, N, 1, device=DEV, generator=g) < 0.5)
    last = torch.randn(N, 1, device=DEV, generator=g) * 0.1
    v = lambda x, d: x.reshape(T, N, d).contiguous()
    return dict(observations=obs, actions=v(act, A), rewards=rew, dones=dones, succs=succs, values=v(val, 1),
                actions_log_prob=v(logp, 1), mu=v(mu, A), sigma=v(ls, A), last_values=last)


def _fill(run, st):
    for tt in range(st["observations"].shape[0]):
        run.storage.add_transitions(st["observations"][tt], st["actions"][tt], st["rewards"][tt, :, 0], st["dones"][tt, :, 0],
                                    st["succs"][tt, :, 0], st["values"][tt], st["actions_log_prob"][tt, :, 0], st["mu"][tt], st["sigma"][tt])


def _bracket(tag, got, o32, f64, sd, lr, o32_b=None, more_draws=None):
    """Per tensor: ||hip - fp64|| / ||fp64 - init|| against BRACKET x the fp32 oracle's distance to fp64 -- its own for that tensor
    or the median over the tensors of the same network, whichever is larger (the distance of a 1- or 32-element bias is one draw of
    a noisy quantity: round 5's first run had the fp32 oracle at 1.3e-4 on `critic.final_mlp.2.bias` between neighbours at 2e-3 ..
    7e-3, the HIP path at 2.7e-3); whole-vector quantiles alike."""
    names = list(sd.keys())
    f64v = np.concatenate([np.asarray(f64[k], dtype=np.float64).reshape(-1) for k in names])
    e_h64, e_o64, e_h32 = per_tensor_update_error(got, f64v, sd), per_tensor_update_error(o32, f64v, sd), per_tensor_update_error(got, o32, sd)
    for extra in (o32_b or []):
        # FURTHER correct fp32 evaluations of the same update (the oracle with its mini-batches cut into pieces of another size: only
        # the summation order differs): the yardstick per tensor is the worst of them -- what fp32 round-off does to THIS trajectory
        # is measured on several draws instead of one (one draw of `critic.sa.2.0.bias` ranged over 5.4e-4 .. 1.09e-3 in six runs)
        e_b = per_tensor_update_error(extra, f64v, sd)
        e_o64 = {k: (max(v[0], e_b[k][0]), v[1]) for k, v in e_o64.items()}
    net_of = lambda k: k.split(".")[0]
    typical = {n: float(np.median([e_o64[k][0] for k in names if k in e_o64 and e_o64[k][1] > 0 and net_of(k) == n]))
               for n in {net_of(k) for k in names if k in e_o64 and e_o64[k][1] > 0}}
    # The HIP side is deterministic, the yardstick is not: a tensor over the bound may just have met a lucky (small) draw of the
    # oracle's own distance.  Before failing, the yardstick is measured on FURTHER fp32 evaluations (`more_draws`: other piece sizes),
    # which can only raise it -- a wrong gradient (10 x and more) stays wrong under any number of draws.  14 runs on the final code
    # with three draws: worst tensor at 0.63 .. 0.92 of the bound, so the further draws are rarely needed.
    over = [k for k in names if k in e_h64 and e_h64[k][1] > 0 and e_h64[k][0] >= BRACKET * max(e_o64[k][0], typical[net_of(k)])]
    if over and more_draws is not None:
        n_extra = 0
        for extra in more_draws():
            e_b = per_tensor_update_error(extra, f64v, sd)
            e_o64 = {k: (max(v[0], e_b[k][0]), v[1]) for k, v in e_o64.items()}
            n_extra += 1
        typical = {n: float(np.median([e_o64[k][0] for k in names if k in e_o64 and e_o64[k][1] > 0 and net_of(k) == n]))
                   for n in {net_of(k) for k in names if k in e_o64 and e_o64[k][1] > 0}}
        record_margin(f"{tag}: further fp32 oracle evaluations drawn because a tensor exceeded the bound on the first draws", n_extra, 4, tensors=over)
    bad = {}
    for k in names:
        if k not in e_h64 or e_h64[k][1] == 0:
            continue
        bound = BRACKET * max(e_o64[k][0], typical[net_of(k)])
        record_margin(f"{tag}: hip-to-fp64 / ({BRACKET} x max(oracle32-to-fp64, its median over the network's tensors)) [{k}]",
                      e_h64[k][0] / bound, 1.0, hip_to_fp64=e_h64[k][0], oracle32_to_fp64=e_o64[k][0], hip_to_oracle32=e_h32[k][0],
                      network_median_oracle32_to_fp64=typical[net_of(k)])
        if e_h64[k][0] >= bound:
            bad[k] = (e_h64[k][0], e_o64[k][0], typical[net_of(k)])
    d_h, d_o = np.abs(got.astype(np.float64) - f64v), np.abs(o32.astype(np.float64) - f64v)
    for name, qq in (("median", 0.5), ("99.9 % quantile", 0.999)):
        a, b = float(np.quantile(d_h, qq)), float(np.quantile(d_o, qq))
        record_margin(f"{tag}: {name} of |param - fp64|, hip / oracle32", a / max(b, 1e-300), BRACKET, hip_over_lr=a / lr, oracle32_over_lr=b / lr)
        assert a < BRACKET * b + 1e-3 * lr, (tag, name, a, b)
    assert not bad, f"{tag}: HIP further from the fp64 trajectory than {BRACKET} x the fp32 oracle: {bad}"
    return max(v[0] for v in e_h32.values() if v[1] > 0)


def _ppo_whole_update(net, seed, tag, geom_check=None, grad_chunk=512, second_oracle_chunk=None):
    from partmanip_amd.algorithms import ppo
    _exact_fp32()
    N, T, O, A, lr = 4096, 8, 3072, 10, 5e-5
    sd = cases.actor_critic_state(net, O, A, 0.5, seed)
    p32 = {k: t(v.copy()).to(DEV) for k, v in sd.items()}
    cfg = _ppo_cfg(net, N, T, DEV, lr)
    obs = _clouds(N, T, seed * 10)
    geom = None
    with tempfile.TemporaryDirectory() as d:
        run = ppo(FakeEnv(N, {"normal_state": O}, A), cfg, FakeLogger(d))
    run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    if geom_check is not None:
        geom = R.pointnet2_geometry(obs.reshape(T * N, O), net)
        geom_check(run, obs.reshape(T * N, O), geom)
    st = _rollout(p32, cfg["model"], obs, seed * 10 + 1, geom)
    ret, adv = R.gae_returns(st["rewards"], st["values"], st["dones"], st["succs"], st["last_values"], 0.99, 0.95, None, False)

    # ---- HIP path: the whole `learn` window (GAE scan + update)
    _fill(run, st)
    run.log_dict = {}
    run.curr_iter = 1
    run.learn(st["last_values"])
    torch.cuda.synchronize()
    assert torch.equal(run.storage.returns, ret), "GAE returns differ from the oracle"
    got, log = flat_state(run.actor_critic.state_dict()), dict(run.log_dict)
    del run
    _free()

    # ---- the oracle on the same tensors: fp32, then fp64
    roll = {k: st[k] for k in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma")}
    roll["returns"], roll["advantages"] = ret, adv
    # (grad_chunk: the restatement evaluates a 2048-cloud mini-batch in pieces -- the same update, tests/test_oracle_golden.py::
    # test_ppo_update_in_row_chunks_equals_one_shot -- because ATen's one-shot fp32 intermediates at this size go wrong on this stack:
    # SparseUNet's gradients came back NaN / off by 2x, while <= 256-cloud pieces agree with fp64 to 1e-6, profiles/HISTORY.md)
    o = R.ppo_update(p32, roll, cfg, 1, geom=geom, grad_chunk=grad_chunk)
    assert len(o["loss_trace"]) == 160 and R.minibatch_size(N * T, 8) == 2048
    _free()
    extra32 = []
    for ch in (second_oracle_chunk or ()):
        pb = {k: t(v.copy()).to(DEV) for k, v in sd.items()}
        R.ppo_update(pb, roll, cfg, 1, geom=geom, grad_chunk=ch)
        extra32.append(flat_state(pb))
        del pb
        _free()
    p64 = {k: t(v.copy()).to(DEV).double() for k, v in sd.items()}
    o64 = R.ppo_update(p64, {k: v.double() for k, v in roll.items()}, cfg, 1, geom=geom, grad_chunk=grad_chunk)
    _free()
    ref, l64 = o["log"], o64["log"]
    assert log["Train/kl_update_count"] == ref["Train/kl_update_count"] == l64["Train/kl_update_count"] == 80
    # KL is a mean of ~1e-3 quantities along a trajectory that two correct fp32 evaluations do not share bit for bit: the fp32 oracle
    # itself moved from 7.8024e-4 to 7.82e-4 (fp64: 7.789e-4) when its mini-batches went from one shot to 512-cloud pieces (cfg 3,
    # round 5); the HIP path sat at 7.80e-4 both times.  Observed |hip - oracle32| / oracle32: 3.2e-3 (cfg 3), 4.6e-5 (PointNet++).
    for k, rtol, atol in (("Train/value_function_loss", 2e-5, 0.0), ("Train/kl", 1e-2, 1e-9), ("Train/kl_max", 1e-2, 1e-9),
                          ("Train/surrogate_loss", 0.0, 2e-5)):
        # against the fp32 oracle the bound is at least BRACKET x the oracle's own distance from fp64 (cfg 3, surrogate loss: the
        # oracle sits 1.3e-5 from fp64, the HIP path 1.9e-6 -- the 2e-5 above would be a test of the ORACLE's rounding there)
        assert_close_rec(f"{tag} {k} vs oracle32", float(log[k]), float(ref[k]), rtol=rtol,
                         atol=max(atol, BRACKET * abs(float(ref[k]) - float(l64[k]))))
        # (the same bound against the fp64 evaluation; a RATIO of the two distances is not recorded: either can be ~0 by chance)
        assert_close_rec(f"{tag} {k} vs fp64", float(log[k]), float(l64[k]), rtol=rtol, atol=atol)
    def more_draws():                                  # only evaluated when a tensor exceeds the bound on the draws above
        for ch in ((96, 160) if second_oracle_chunk else ()):
            pb = {k: t(v.copy()).to(DEV) for k, v in sd.items()}
            R.ppo_update(pb, roll, cfg, 1, geom=geom, grad_chunk=ch)
            yield flat_state(pb)
            del pb
            _free()

    worst = _bracket(tag, got, flat_state(p32), {k: v.cpu().numpy() for k, v in p64.items()}, sd, lr,
                     o32_b=extra32, more_draws=more_draws)
    print(f"{tag}: worst per-tensor ||hip - oracle32|| / ||oracle32 - init|| = {worst:.2e}")


def test_cfg3_whole_iteration_matches_oracle_fp32_and_fp64():
    """(a): BASELINE cfg 3 -- the bench's headline workload `ppo_vision_pointnet_4096env_x_8step_x_1024pt` -- GAE + 160 optimiser steps."""
    _ppo_whole_update(dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False), 841, "cfg 3 whole iteration")


def test_vision_pn2_whole_iteration_matches_oracle_fp32_and_fp64():
    """(b): `ppo_vision_pointnet2ssg_4096env_x_8step_x_1024pt` -- and the neighbourhood tables of the whole rollout, bit for bit."""
    net = dict(name="PointNet2", activation="tanh")

    def tables_equal(run, flat_obs, geom):
        tabs = run.actor_critic.actor.precompute_geometry(flat_obs)
        xyz = flat_obs.view(-1, 1024, 3)
        for l, ((centers, idx), (idx_c, idx_g)) in enumerate(zip(tabs, geom)):
            want = torch.gather(xyz, 1, idx_c.unsqueeze(-1).expand(-1, -1, 3))
            assert torch.equal(centers, want), f"level {l}: FPS centres differ from the restatement"
            assert torch.equal(idx.long(), idx_g), f"level {l}: ball-query tables differ from the restatement"
            xyz = want
        record_margin("vision_pn2: FPS centres + ball-query tables of 32768 clouds vs restatement (mismatches)", 0, 0)
    _ppo_whole_update(net, 842, "vision_pn2 whole iteration", geom_check=tables_equal, grad_chunk=256, second_oracle_chunk=(128, 192))


def test_dagger_sparse_unet_update_at_2048_clouds_matches_oracle_fp32_and_fp64(tmp_path, monkeypatch):
    """(c): ring of 4096 rows x 4096 voxels (x, y, z, tsdf), n_minibatches 2 -> 2048-cloud mini-batches (the bench's), 2 epochs, random
    sampler (the same `torch.randperm` draws on both sides), frozen state-MLP teacher (O = 53)."""
    from partmanip_amd.algorithms import ppo, dagger
    from partmanip_amd.feeder import FeederEnv
    _exact_fp32()
    monkeypatch.chdir(tmp_path)
    P, A, N, buf, O_t, lr = 4096, 10, 1024, 4, 53, 5e-5
    O_s = 4 * P
    net = dict(name="SparseUNet", activation="tanh", point_num=P, grid=50)
    tnet = dict(name="MLP", hid_dim=[512, 512, 512], activation="tanh")
    model = lambda n, std: dict(action_std=std, action_activate="tanh", clipAction=1.0, network=dict(n))
    tcfg = _ppo_cfg(tnet, N, 1, DEV)
    tea = ppo(FakeEnv(N, {"normal_state": O_t}, A), tcfg, FakeLogger(str(tmp_path)))
    tsd = cases.actor_critic_state(tnet, O_t, A, 0.5, 852)
    tea.actor_critic.load_state_dict({k: t(v.copy()) for k, v in tsd.items()})
    tea.save(1)
    del tea
    cfg = dict(num_envs=N, obs_mode="depth_sparse", model=model(net, 0.1), max_iterations=10000, n_steps=1, n_updates=2,
               n_minibatches=2, device=DEV, buf_size=buf, reward_reset=False, add_proprio_obs=False, offline_data_pth=None,
               eval_round=1, eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
               lr_schedule="fixed", lr=lr, teacher=str(tmp_path / "model_1.pth"), resume=None, pretrain=None, sampler="random")
    env = FeederEnv(N, {"normal_state": O_t, "depth_sparse": O_s, "proprio_state": 0}, A, DEV, seed=853, point_num=P)
    run = dagger(env, cfg, FakeLogger(str(tmp_path)))
    init = cases.actor_critic_state(net, O_s, A, 0.1, 851)
    run.student.load_state_dict({k: t(v.copy()) for k, v in init.items()})
    for _ in range(buf):
        obs = env.reset()
        run.storage.add_transitions_dagger(obs["depth_sparse"], obs["normal_state"])
    ring_obs, ring_tea = run.storage.observations.view(-1, O_s).clone(), run.storage.tea_obs.view(-1, O_t).clone()
    assert run.storage.cur_buf_size == N * buf == 4096
    torch.manual_seed(8530)
    run.log_dict = {}
    run.update(1)
    torch.cuda.synchronize()
    got, loss = flat_state(run.student.state_dict()), float(run.log_dict["Train/dagger_loss"])
    del run
    _free()

    ocfg = dict(model=model(net, 0.1), tea_model=model(tnet, 0.5), n_updates=2, n_minibatches=2, sampler="random", lr=lr,
                lr_schedule="fixed", max_iterations=10000, proprio_shape=0)
    tea32 = {k: t(v.copy()).to(DEV) for k, v in tsd.items()}
    stu32 = {k: t(v.copy()).to(DEV) for k, v in init.items()}
    torch.manual_seed(8530)
    o = R.dagger_update(stu32, tea32, ring_obs, ring_tea, N * buf, ocfg, 1, grad_chunk=256)
    assert len(o["loss_trace"]) == 4 and R.minibatch_size(N * buf, 2) == 2048
    _free()
    stu64 = {k: t(v.copy()).to(DEV).double() for k, v in init.items()}
    torch.manual_seed(8530)
    o64 = R.dagger_update(stu64, {k: v.double() for k, v in tea32.items()}, ring_obs.double(), ring_tea.double(), N * buf, ocfg, 1, grad_chunk=256)
    _free()
    assert_close_rec("dagger SparseUNet 2048 clouds: Train/dagger_loss vs oracle32", loss, o["log"]["Train/dagger_loss"], rtol=2e-5)
    assert_close_rec("dagger SparseUNet 2048 clouds: Train/dagger_loss vs fp64", loss, o64["log"]["Train/dagger_loss"], rtol=2e-5)
    # dagger.py:56: one Adam over every student parameter; only the actor receives gradients -- the critic and log_std must not move
    moved = {k: v for k, v in init.items() if k.startswith("actor.")}
    names = list(init.keys())
    sel = np.concatenate([np.full(int(np.asarray(init[k]).size), k.startswith("actor.")) for k in names])
    init_flat = np.concatenate([np.asarray(v, dtype=np.float32).reshape(-1) for v in init.values()])
    assert np.array_equal(got[~sel], init_flat[~sel]), "critic / log_std moved"
    worst = _bracket("dagger SparseUNet 2048 clouds", got[sel], flat_state(stu32)[sel], {k: stu64[k].cpu().numpy() for k in moved}, moved, lr)
    print(f"dagger SparseUNet: worst per-tensor ||hip - oracle32|| / ||oracle32 - init|| = {worst:.2e}")
