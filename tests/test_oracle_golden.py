"""Pin the CPU oracle (oracle/ref_cpu.py) to the reference: every golden vector
captured by running the reference itself (tests/golden/make_golden.py) must be
reproduced.  CPU-only; no GPU, no /root/reference at run time."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.golden import cases
from tests.helpers import load_fixture, t, state_dict_t, ppo_cfg, ppo_rollout, flat_state, assert_params_close


@pytest.mark.parametrize("name", list(cases.GAE_CASES))
def test_gae_bit_exact(name):
    c, fx = cases.GAE_CASES[name], load_fixture(name)
    inp = cases.gae_inputs(c)
    ret, adv = R.gae_returns(t(inp["rewards"]), t(inp["values"]), t(inp["dones"]), t(inp["succs"]),
                             t(inp["last_values"]), 0.99, 0.95, c["succ_value"], c["whole_adv_norm"])
    assert np.array_equal(ret.numpy(), fx["returns"])
    assert np.array_equal(adv.numpy(), fx["advantages"])


@pytest.mark.parametrize("name", list(cases.PPO_CASES))
def test_actor_critic_forward(name):
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    p = state_dict_t(cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"]))
    st = ppo_rollout(c, fx)
    cfg = ppo_cfg(c)
    with torch.no_grad():
        logp, ent, val, mu, ls = R.update_act_cri(p, cfg["model"], st["observations"].view(-1, c["O"]),
                                                  st["actions"].view(-1, c["A"]))
    np.testing.assert_allclose(mu.numpy(), fx["fwd_mu"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(val.numpy(), fx["fwd_value"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(logp.numpy(), fx["fwd_logp"], rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(ent.numpy(), fx["fwd_entropy"], rtol=1e-6)


@pytest.mark.parametrize("name", list(cases.PPO_CASES))
def test_gae_inside_ppo_cases(name):
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    st = ppo_rollout(c, fx)
    ret, adv = R.gae_returns(st["rewards"], st["values"], st["dones"], st["succs"], st["last_values"],
                             c["gamma"], c["lam"], c["succ_value"], c["tricks"]["whole_adv_norm"])
    assert np.array_equal(ret.numpy(), fx["returns"])
    assert np.array_equal(adv.numpy(), fx["advantages"])


@pytest.mark.parametrize("name", list(cases.PPO_CASES))
def test_sampler_lists(name):
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    n = c["T"] * c["N"]
    if c["sampler"] == "random":
        torch.manual_seed(c["seed"])
        got = [R.minibatch_index_lists(n, c["n_minibatches"], "random") for _ in range(2 * c["n_updates"])]
    else:
        got = [R.minibatch_index_lists(n, c["n_minibatches"], "sequential")]
    assert np.array_equal(np.array(got, dtype=np.int64), fx["index_lists"])


@pytest.mark.parametrize("name", list(cases.PPO_CASES))
def test_ppo_update(name):
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    p = state_dict_t(cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"]))
    st = ppo_rollout(c, fx)
    st["returns"], st["advantages"] = t(fx["returns"]), t(fx["advantages"])
    cfg = ppo_cfg(c)
    if c["sampler"] == "random":
        torch.manual_seed(c["seed"])
    out = R.ppo_update(p, {k: st[k] for k in ("observations", "actions", "values", "returns", "actions_log_prob",
                                             "advantages", "mu", "sigma")}, cfg, c["it"])
    np.testing.assert_allclose(out["loss_trace"], fx["loss_trace"], rtol=2e-4, atol=2e-6)
    log = out["log"]
    assert log["Train/kl_update_count"] == int(fx["log_kl_update_count"])
    for k in ("value_function_loss", "surrogate_loss", "kl", "kl_max", "learning_rate",
              "value_gt_return_mean", "value_gt_return_max"):
        np.testing.assert_allclose(log["Train/" + k], float(fx["log_" + k]), rtol=2e-4, atol=2e-6, err_msg=k)
    fin = flat_state(p)
    s = int(fx["final_stride"])
    # same torch, different host CPU => different sgemm summation order; Adam turns a gradient element that is ~0
    # at step 1 into a move of up to lr whatever its sign, so a handful of elements may differ by O(lr):
    # 99.9 % within 2e-5, none beyond 2.5 * lr * steps (the bound the GPU tests use, tests/test_gpu_learner.py)
    diff = np.abs(fin[::s].astype(np.float64) - fx["final_flat"].astype(np.float64))
    assert np.quantile(diff, 0.999) < 2e-5, np.quantile(diff, 0.999)
    assert diff.max() < 2.5 * c["lr"] * len(fx["loss_trace"]), diff.max()
    np.testing.assert_allclose(fin.astype(np.float64).sum(), float(fx["final_sum"]), rtol=1e-6, atol=1e-3)
    adam_a = out["opt"][0]
    np.testing.assert_allclose(adam_a.m[-1].numpy(), fx["adam_logstd_m"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(adam_a.v[-1].numpy(), fx["adam_logstd_v"], rtol=1e-3, atol=1e-9)
    assert adam_a.t[-1] == int(fx["adam_step"])
    np.testing.assert_allclose([adam_a.lrs[0], adam_a.lrs[-1]], fx["lr_actor_groups"])
    np.testing.assert_allclose([out["opt"][1].lrs[0]], fx["lr_critic_groups"])


@pytest.mark.parametrize("name", list(cases.DAGGER_CASES))
def test_dagger(name):
    c, fx = cases.DAGGER_CASES[name], load_fixture(name)
    A = c["A"]
    stu = state_dict_t(cases.actor_critic_state(c["stu_net"], c["O_s"], A, c["action_std"], c["seed"], c["proprio"]))
    tea = state_dict_t(cases.actor_critic_state(c["tea_net"], c["O_t"], A, 0.5, c["seed"] + 1))
    raw = cases.dagger_raw_inputs(c)
    cap = c["buf_size"] * c["N"]
    ring_obs, ring_tea = torch.zeros(cap, c["O_s"]), torch.zeros(cap, c["O_t"])
    ind, size = 0, 0
    for k in range(c["n_fill"]):
        ind, size = R.dagger_ring_insert(ring_obs, ring_tea, ind, size, t(raw["stu"][k]), t(raw["tea"][k]))
    assert (ind, size) == (int(fx["mix_buf_ind"]), int(fx["cur_buf_size"]))
    assert np.array_equal(ring_tea.numpy(), fx["ring_tea"])
    np.testing.assert_allclose(float(ring_obs.double().sum()), float(fx["ring_obs_sum"]), rtol=1e-12)
    stu_model = dict(action_std=c["action_std"], action_activate="tanh", clipAction=1.0, network=dict(c["stu_net"]))
    tea_model = dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=dict(c["tea_net"]))
    with torch.no_grad():
        np.testing.assert_allclose(R.act(tea, tea_model, ring_tea).numpy(), fx["tea_act"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(R.act(stu, stu_model, ring_obs, c["proprio"]).numpy(), fx["stu_act0"],
                                   rtol=1e-5, atol=2e-6)
    torch.manual_seed(c["torch_seed"])
    lists = [R.minibatch_index_lists(size, c["n_minibatches"], c["sampler"]) for _ in range(c["n_updates"])]
    assert np.array_equal(np.array(lists, dtype=np.int64), fx["index_lists"])
    torch.manual_seed(c["torch_seed"])
    cfg = dict(model=stu_model, tea_model=tea_model, n_updates=c["n_updates"], n_minibatches=c["n_minibatches"],
               sampler=c["sampler"], lr=c["lr"], lr_schedule=c["lr_schedule"], max_iterations=c["max_iterations"],
               proprio_shape=c["proprio"])
    out = R.dagger_update(stu, tea, ring_obs, ring_tea, size, cfg, c["it"])
    np.testing.assert_allclose(out["loss_trace"], fx["loss_trace"], rtol=2e-4, atol=1e-8)
    np.testing.assert_allclose(out["log"]["Train/dagger_loss"], float(fx["log_dagger_loss"]), rtol=2e-4)
    np.testing.assert_allclose(out["log"]["Train/learning_rate"], float(fx["log_learning_rate"]), rtol=1e-12)
    fin = flat_state(stu)
    s = int(fx["final_stride"])
    assert_params_close(fin[::s], fx["final_flat"], c["lr"], len(fx["loss_trace"]))


def test_dagger_resume_and_load_pretrain_match_reference():
    """dagger.py:98-120 by the reference itself (tests/golden/make_golden.py::gen_dagger_ckpt): a student checkpoint the reference
    wrote after one update (ref_ckpt_dagger_mlp.pth), the reference resuming from it and taking a second update, and
    `load_pretrain` into another student.  The restatement continues from the same checkpoint and must land where the reference
    did; the checkpoint's layout is the one A14 describes (one Adam over all student parameters, state only where a gradient
    arrived: the actor's tensors)."""
    import os
    from tests.helpers import GOLDEN
    c, fx = cases.DAGGER_CASES["dagger_mlp"], load_fixture("dagger_mlp_ckpt")
    ck = torch.load(os.path.join(GOLDEN, "ref_ckpt_dagger_mlp.pth"), map_location="cpu", weights_only=False)
    assert set(ck) == {"iteration", "model_state_dict", "optimizer_state_dict", "total_steps", "obs_mode", "teacher"}
    assert ck["iteration"] == c["it"] and ck["total_steps"] == 4321 and ck["obs_mode"] == "stu_mode"
    names = list(ck["model_state_dict"].keys())
    assert names[0] == "log_std" and np.array_equal(flat_state(ck["model_state_dict"]), fx["saved_flat"])
    n_actor = sum(1 for k in names if k.startswith("actor."))
    osd = ck["optimizer_state_dict"]
    assert sorted(osd["state"].keys()) == list(range(1, 1 + n_actor)) and len(osd["param_groups"]) == 1
    stu = {k: v.clone() for k, v in ck["model_state_dict"].items()}
    tea = state_dict_t(cases.actor_critic_state(c["tea_net"], c["O_t"], c["A"], 0.5, c["seed"] + 1))
    opt = R.Adam([stu[k] for k in names], osd["param_groups"][0]["lr"])
    for i, st_ in osd["state"].items():
        opt.m[i], opt.v[i], opt.t[i] = st_["exp_avg"].clone(), st_["exp_avg_sq"].clone(), int(st_["step"])
    raw = cases.dagger_raw_inputs(c)
    cap = c["buf_size"] * c["N"]
    ring_obs, ring_tea = torch.zeros(cap, c["O_s"]), torch.zeros(cap, c["O_t"])
    ind, size = 0, 0
    for k in range(c["n_fill"]):
        ind, size = R.dagger_ring_insert(ring_obs, ring_tea, ind, size, t(raw["stu"][k]), t(raw["tea"][k]))
    model = lambda net, std: dict(action_std=std, action_activate="tanh", clipAction=1.0, network=dict(net))
    cfg = dict(model=model(c["stu_net"], c["action_std"]), tea_model=model(c["tea_net"], 0.5), n_updates=c["n_updates"],
               n_minibatches=c["n_minibatches"], sampler=c["sampler"], lr=c["lr"], lr_schedule=c["lr_schedule"],
               max_iterations=c["max_iterations"], proprio_shape=c["proprio"])
    torch.manual_seed(c["torch_seed"] + 1)
    out = R.dagger_update(stu, tea, ring_obs, ring_tea, size, cfg, c["it"] + 1, opt=opt)
    np.testing.assert_allclose(out["loss_trace"], fx["resume_loss_trace"], rtol=2e-4, atol=1e-8)
    np.testing.assert_allclose(out["log"]["Train/learning_rate"], float(fx["resume_log_learning_rate"]), rtol=1e-12)
    assert_params_close(flat_state(stu), fx["resume_final_flat"], c["lr"], len(fx["resume_loss_trace"]))
    assert [opt.t[i] for i in range(1, 1 + n_actor)] == [int(x) for x in fx["resume_adam_steps"]]
    # load_pretrain: everything but log_std comes from the checkpoint (dagger.py:102-103)
    other = cases.actor_critic_state(c["stu_net"], c["O_s"], c["A"], 0.3, c["seed"] + 50, c["proprio"])
    want = np.concatenate([np.asarray(other["log_std"], np.float32).reshape(-1), fx["saved_flat"][other["log_std"].size:]])
    assert np.array_equal(fx["pretrain_flat"], want) and np.array_equal(fx["pretrain_log_std"], other["log_std"])


@pytest.mark.parametrize("name", ["ppo_mlp_allon", "ppo_mlp_klskip", "ppo_mlp_random"])
def test_ppo_update_in_row_chunks_equals_one_shot(name):
    """`grad_chunk` (what the whole-update GPU tests evaluate the restatement with: a 2048-cloud mini-batch in pieces) is the same
    update: in fp64 the chunked and the one-shot evaluation agree to round-off -- with mini-batch advantage normalisation and the
    clipped value loss on (batch statistics over the WHOLE mini-batch), with KL skips, with the random sampler."""
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    st = ppo_rollout(c, fx)
    st["returns"], st["advantages"] = t(fx["returns"]), t(fx["advantages"])
    keys = ("observations", "actions", "values", "returns", "actions_log_prob", "advantages", "mu", "sigma")
    res = []
    for chunk in (None, 5):
        p = {k: v.double() for k, v in state_dict_t(cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"])).items()}
        torch.manual_seed(c["seed"])
        out = R.ppo_update(p, {k: st[k].double() for k in keys}, ppo_cfg(c), c["it"], grad_chunk=chunk)
        res.append((np.concatenate([v.numpy().reshape(-1) for v in p.values()]), out))
    (pa, oa), (pb, ob) = res
    np.testing.assert_allclose(pb, pa, rtol=0, atol=1e-9)
    assert oa["log"]["Train/kl_update_count"] == ob["log"]["Train/kl_update_count"]
    np.testing.assert_allclose(ob["loss_trace"], oa["loss_trace"], rtol=1e-10, atol=1e-14)
    for k in ("Train/surrogate_loss", "Train/kl", "Train/kl_max", "Train/value_function_loss"):
        np.testing.assert_allclose(ob["log"][k], oa["log"][k], rtol=1e-10, atol=1e-14, err_msg=k)


def test_dagger_update_in_row_chunks_equals_one_shot():
    c = cases.DAGGER_CASES["dagger_mlp"]
    tea = {k: v.double() for k, v in state_dict_t(cases.actor_critic_state(c["tea_net"], c["O_t"], c["A"], 0.5, c["seed"] + 1)).items()}
    raw = cases.dagger_raw_inputs(c)
    cap = c["buf_size"] * c["N"]
    ring_obs, ring_tea = torch.zeros(cap, c["O_s"]), torch.zeros(cap, c["O_t"])
    ind, size = 0, 0
    for k in range(c["n_fill"]):
        ind, size = R.dagger_ring_insert(ring_obs, ring_tea, ind, size, t(raw["stu"][k]), t(raw["tea"][k]))
    model = lambda net, std: dict(action_std=std, action_activate="tanh", clipAction=1.0, network=dict(net))
    cfg = dict(model=model(c["stu_net"], c["action_std"]), tea_model=model(c["tea_net"], 0.5), n_updates=c["n_updates"],
               n_minibatches=c["n_minibatches"], sampler=c["sampler"], lr=c["lr"], lr_schedule=c["lr_schedule"],
               max_iterations=c["max_iterations"], proprio_shape=c["proprio"])
    res = []
    for chunk in (None, 7):
        stu = {k: v.double() for k, v in state_dict_t(cases.actor_critic_state(c["stu_net"], c["O_s"], c["A"], c["action_std"], c["seed"])).items()}
        torch.manual_seed(c["torch_seed"])
        out = R.dagger_update(stu, tea, ring_obs.double(), ring_tea.double(), size, cfg, c["it"], grad_chunk=chunk)
        res.append((np.concatenate([v.numpy().reshape(-1) for v in stu.values()]), out["loss_trace"]))
    np.testing.assert_allclose(res[1][1], res[0][1], rtol=1e-10)
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=0, atol=1e-9)


def test_slab_evaluated_linear_equals_autograd_linear():
    """`_LinearSlabs` (the restatement's Linear for >= 2^16 rows on a device: weight gradient as a batched product over slabs of the
    rows) against autograd's F.linear in fp64: output identical, gradients to round-off -- 2-D and 3-D inputs, a ragged row count."""
    g = torch.Generator().manual_seed(0)
    for shape in ((1000, 12), (7, 300, 5)):
        x = torch.randn(*shape, generator=g, dtype=torch.float64, requires_grad=True)
        w = torch.randn(9, shape[-1], generator=g, dtype=torch.float64, requires_grad=True)
        b = torch.randn(9, generator=g, dtype=torch.float64, requires_grad=True)
        u = torch.randn(*shape[:-1], 9, generator=g, dtype=torch.float64)
        ya = torch.nn.functional.linear(x, w, b)
        ga = torch.autograd.grad((ya * u).sum(), (x, w, b))
        yb = R._LinearSlabs.apply(x, w, b)
        gb = torch.autograd.grad((yb * u).sum(), (x, w, b))
        assert torch.equal(ya, yb)
        for a, c in zip(ga, gb):
            np.testing.assert_allclose(c.numpy(), a.numpy(), rtol=1e-12, atol=1e-12)


def test_dagger_small_buffer_returns_early():
    assert R.dagger_update({}, {}, None, None, 15, {}, 1) is None


def test_depth2pc_restatement_matches_reference_world_cloud():
    """utils/depth2tsdf.py:136-157 run by the REFERENCE's own code (make_golden.gen_depth2pc): the restatement's
    world cloud (back-projection + pose + crop) is bit-identical; the sampling after it has no reference
    implementation here (pytorch3d absent) -- the fixture's indices are the restatement's own (parity unpinned)."""
    import numpy as np
    from tests.golden import cases
    from tests.helpers import load_fixture
    from oracle import ref_cpu as R
    c = cases.DEPTH2PC_CASES["depth2pc_small"]
    inp, fx = cases.depth2pc_inputs(c), load_fixture("depth2pc_small")
    out, world, idx = R.depth2pc(inp["depth"], inp["cam_pose"], c["intr"], c["size"], c["vol_origin"], K=c["K"],
                                 return_world=True)
    assert np.array_equal(world, fx["world"])
    assert 0.2 < float((world != 0).any(-1).mean()) < 0.8           # the crop is exercised both ways
    assert np.array_equal(idx, fx["idx"])
    assert np.array_equal(out, fx["final_pc_1024"][:, :c["K"]])     # reference gather of those indices


def test_bc_restatement_matches_reference_run():
    """algorithms/bc.py `bc.run()` executed by the reference itself (10 DataLoader workers, shards on disk): the
    restatement reproduces its shuffled index batches from the same global RNG state, the per-iteration mean
    losses, the lr schedule and the final student."""
    import numpy as np
    import torch
    from tests.golden import cases
    from tests.helpers import load_fixture, flat_state
    from oracle import ref_cpu as R
    c, fx = cases.BC_CASES["bc_mlp"], load_fixture("bc_mlp")
    d = {k: torch.from_numpy(v) for k, v in cases.bc_dataset(c).items()}
    sd = cases.actor_critic_state(c["net"], c["D"] + c["S"], c["A"], c["action_std"], c["seed"])
    p = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    torch.manual_seed(c["torch_seed"])
    out = R.bc_run(p, d, dict(lr=c["lr"], lr_schedule=c["lr_schedule"], max_iterations=c["max_iterations"],
                              n_minibatches=c["n_minibatches"], add_proprio_obs=True, action_activate="tanh",
                              max_action=1.0), c["net"])
    np.testing.assert_allclose(out["loss_trace"], fx["loss_trace"], rtol=2e-6)
    np.testing.assert_allclose(out["lr_trace"], fx["lr_trace"], rtol=1e-12)
    np.testing.assert_allclose(flat_state(p), fx["final_flat"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["conv3d_proprio", "conv3d_plain"])
def test_conv3dnet_restatement_matches_reference_module(name):
    """network.py:67-94 `Conv3DNet` instantiated and run by the reference itself (make_golden.gen_conv3d): outputs
    and a strided sample of every parameter gradient."""
    import numpy as np
    import torch
    from tests.golden import cases
    from tests.helpers import load_fixture
    from oracle import ref_cpu as R
    c, fx = cases.CONV3D_CASES[name], load_fixture(name)
    p = {"actor." + k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in cases.conv3d_state(c).items()}
    inp = cases.conv3d_inputs(c)
    out = R.net_forward(p, "actor", dict(name="Conv3DNet", activation="tanh"), torch.from_numpy(inp["x"]), c["proprio"])
    np.testing.assert_allclose(out.detach().numpy(), fx["out"], rtol=1e-5, atol=1e-6)
    grads = torch.autograd.grad((out * torch.from_numpy(inp["dy"])).sum(), list(p.values()))
    for k, g in zip(p, grads):
        ref = fx["grad_" + k[len("actor."):]]
        np.testing.assert_allclose(g.numpy().reshape(-1)[::7], ref, rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(ref).max())))


def test_tsdf_integrate_restatement_matches_reference():
    """utils/depth2tsdf.py:68-86 run by the reference's own TSDFVolume (resolution 10): bit-identical volume."""
    import numpy as np
    from tests.golden import cases
    from tests.helpers import load_fixture
    from oracle import ref_cpu as R
    c = cases.DEPTH2PC_CASES["depth2pc_small"]
    inp, fx = cases.depth2pc_inputs(c), load_fixture("depth2pc_small")
    import torch
    idx, z = R.tsdf_tables(inp["cam_pose"], c["intr"], c["h"], c["w"], c["size"], 10, c["vol_origin"])
    # the tables come out of a host bmm: pixel indices must agree, depths to the last bit or two (CPU-dependent)
    assert np.array_equal(idx.numpy(), fx["tsdf_pix_idx"])
    np.testing.assert_allclose(z.numpy(), fx["tsdf_pix_z"], rtol=3e-7)
    vol = R.tsdf_integrate(inp["depth"], torch.from_numpy(fx["tsdf_pix_idx"]).long(), torch.from_numpy(fx["tsdf_pix_z"]),
                           c["size"], 10).numpy()
    assert np.array_equal(vol, fx["tsdf"])                          # with the reference's tables: bit-identical
    assert 0.05 < float((vol != 1.0).mean()) < 0.95                 # some voxels are seen, some keep the default


def test_sparse_voxel_restatement_matches_reference():
    """utils/depth2tsdf.py:88-120 run by the reference's own TSDFVolume.sparse_voxel (its pytorch3d dependency stubbed
    by the restated FPS, make_golden.gen_depth2pc): (b, 1024, 4) rows incl. the zero-padded tail, bit-identical."""
    fx = load_fixture("depth2pc_small")
    out = R.tsdf_sparse_voxel(fx["tsdf"], K=1024).numpy()
    assert out.dtype == np.float32 and np.array_equal(out, fx["sparse_voxel"])
    n_band = ((fx["tsdf"] < 0.2) & (fx["tsdf"] > -0.2)).reshape(2, -1).sum(-1)
    assert (n_band < 1024).all() and (n_band > 100).all()           # the fixture exercises the padded branch
    for b in range(2):                                               # padding rows read voxel (0,0,0)
        assert np.array_equal(out[b, n_band[b]:, :3], np.zeros((1024 - n_band[b], 3), np.float32))
        assert (out[b, n_band[b]:, 3] == fx["tsdf"][b, 0, 0, 0]).all()


def _rollout_inputs():
    from tests.golden.detgen import det_uniform, det_normal
    c = cases.ROLLOUT_CASE
    xs = [(det_normal((c["N"], c["O"]), c["seed"] + i) * (1.0 + 0.5 * i) + det_uniform((1, c["O"]), c["seed"] + 9, -2, 2))
          .astype(np.float32) for i in range(3)]
    obs = det_normal((c["N"], c["O"]), c["seed"] + 20).astype(np.float32)
    return c, xs, obs


def test_rollout_side_restatement_matches_reference():
    """RMS.py:10-18,40-45 and actor_critic.py:36-47 run by the reference's own `Normalization` / `ActorCritic`
    (make_golden.gen_rollout): running statistics after each of three batches, the normalised batches, the frozen
    (update=False) call, and `random_act_cri` with the recorded standard-normal draw (identical on the generating
    host; to the last bit or two on any other CPU)."""
    c, xs, obs = _rollout_inputs()
    fx = load_fixture("rollout_side")
    rms = R.RunningMeanStd(c["O"])
    for i, x in enumerate(xs):
        out = rms.normalize(t(x))
        # torch's column reductions round differently on different host CPUs (vector width): last-bit tolerance, not ==
        np.testing.assert_allclose(out.numpy(), fx["norm_out"][i], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(np.stack([rms.mean.numpy()[0], rms.std.numpy()[0], rms.S.numpy()[0]]),
                                   fx["norm_stats"][i], rtol=2e-6, atol=1e-7)
    assert rms.n == int(fx["n"])
    np.testing.assert_allclose(rms.normalize(t(xs[2]), update=False).numpy(), fx["norm_frozen"], rtol=2e-6, atol=2e-6)
    p = state_dict_t(cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"]))
    model = dict(action_std=c["action_std"], action_activate="tanh", clipAction=c["max_action"], network=dict(c["net"]))
    with torch.no_grad():
        act, logp, val, mu, ls = R.random_act_cri(p, model, t(obs), t(fx["eps"]))
    np.testing.assert_allclose(mu.numpy(), fx["mu"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(val.numpy(), fx["value"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(act.numpy(), fx["actions"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(logp.numpy(), fx["logp"], rtol=2e-5, atol=1e-4)
    assert np.array_equal(ls.numpy(), fx["log_std_rows"])


def test_dagger_update_on_offline_plus_on_policy_ring_matches_reference():
    """`dagger.update` over the ring the reference built from offline shards + on-policy rows (dagger_offline.npz)."""
    c, fx = cases.DAGGER_OFFLINE_CASE, load_fixture("dagger_offline")
    A, O_s = c["A"], c["D"] + c["proprio"]
    stu = state_dict_t(cases.actor_critic_state(c["stu_net"], O_s, A, c["action_std"], c["seed"]))
    tea = state_dict_t(cases.actor_critic_state(c["tea_net"], c["O_t"], A, 0.5, c["seed"] + 1))
    model = lambda net, std: dict(action_std=std, action_activate="tanh", clipAction=1.0, network=dict(net))
    cfg = dict(model=model(c["stu_net"], c["action_std"]), tea_model=model(c["tea_net"], 0.5), n_updates=c["n_updates"],
               n_minibatches=c["n_minibatches"], sampler=c["sampler"], lr=c["lr"], lr_schedule=c["lr_schedule"],
               max_iterations=c["max_iterations"], proprio_shape=0)
    torch.manual_seed(c["torch_seed"])
    out = R.dagger_update(stu, tea, t(fx["ring_obs"]), t(fx["ring_tea"]), int(fx["state"][1]), cfg, c["it"])
    np.testing.assert_allclose(out["loss_trace"], fx["loss_trace"], rtol=2e-4, atol=1e-8)
    np.testing.assert_allclose(out["log"]["Train/dagger_loss"], float(fx["log_dagger_loss"]), rtol=2e-4)
    np.testing.assert_allclose(out["log"]["Train/learning_rate"], float(fx["log_learning_rate"]), rtol=1e-12)
    assert_params_close(flat_state(stu), fx["final_flat"], c["lr"], len(fx["loss_trace"]))
