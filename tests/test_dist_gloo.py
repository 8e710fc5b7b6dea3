"""N > 1 path on CPU: world_size-2 gloo process groups exercise partmanip_amd.dist (env sharding,
the single flat all-reduce per optimiser step with the step scalars in its tail, the moments
reduction) -- with the CPU oracle standing in for the per-rank compute, which is the only role the
oracle may play.  Parity definition: W ranks x mini-batch b with mean-reduced
gradients == one process with mini-batch W*b over the union of the ranks' rows."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.golden import cases
from tests.helpers import load_fixture, ppo_cfg, ppo_rollout, state_dict_t, t, flat_state, assert_params_close


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleSyncAdapter:
    """Adapts partmanip_amd.dist.GradSync to the two hooks oracle.ppo_update calls."""

    def __init__(self, sync):
        self.sync = sync

    def mean_grads(self, grads):
        flat = torch.cat([g.reshape(-1) for g in grads])          # ONE message per optimiser step
        self.sync.mean_(flat)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def mean_scalar(self, x):
        return self.sync.mean_(x.clone().reshape(1))[0]


def _rank_main(rank, world, port, name, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import ref_cpu as R
    from partmanip_amd import dist as pdist
    r, w, _ = pdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    sync = pdist.maybe_sync()
    assert sync is not None and sync.world == world
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    st = ppo_rollout(c, fx)
    st["returns"], st["advantages"] = t(fx["returns"]), t(fx["advantages"])
    lo, hi = pdist.shard_envs(c["N"], rank, world)
    keys = ("observations", "actions", "values", "returns", "actions_log_prob", "advantages", "mu", "sigma")
    local = {k: st[k][:, lo:hi].contiguous() for k in keys}        # this rank's (T, N/W, .) storage
    p = state_dict_t(cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"]))
    out = R.ppo_update(p, local, ppo_cfg(c), c["it"], grad_sync=OracleSyncAdapter(sync))
    # moments reduction used by whole_adv_norm / mini_adv_norm under data parallelism
    adv = local["advantages"].double().reshape(-1)
    mom = torch.stack([adv.sum(), (adv * adv).sum()])
    count = sync.moments_sync(mom, adv.numel())
    np.save(os.path.join(out_dir, f"r{rank}.npy"), flat_state(p))
    np.save(os.path.join(out_dir, f"m{rank}.npy"), np.array([float(mom[0]), float(mom[1]), count]))
    np.save(os.path.join(out_dir, f"c{rank}.npy"), np.array([out["log"]["Train/kl_update_count"]]))
    np.save(os.path.join(out_dir, f"a{rank}.npy"), np.array([float(sync.any_(rank == 3 % world)), float(sync.any_(False))]))
    sync.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("name", ["ppo_mlp_default", "ppo_mlp_klskip"])
def test_two_rank_dp_equals_single_process(name, tmp_path):
    from oracle import ref_cpu as R
    world, port = 2, _free_port()
    mp.spawn(_rank_main, args=(world, port, name, str(tmp_path)), nprocs=world, join=True)
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    st = ppo_rollout(c, fx)
    st["returns"], st["advantages"] = t(fx["returns"]), t(fx["advantages"])
    keys = ("observations", "actions", "values", "returns", "actions_log_prob", "advantages", "mu", "sigma")
    p = state_dict_t(cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"]))
    ref = R.ppo_update(p, {k: st[k] for k in keys}, ppo_cfg(c), c["it"])     # mini-batch = W * b, same row sets
    want = flat_state(p)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1), "ranks diverged"
    assert_params_close(r0, want, c["lr"], len(ref["loss_trace"]))
    assert int(np.load(tmp_path / "c0.npy")[0]) == ref["log"]["Train/kl_update_count"]
    adv = st["advantages"].double().reshape(-1)
    m0 = np.load(tmp_path / "m0.npy")
    np.testing.assert_allclose(m0[:2], [float(adv.sum()), float((adv * adv).sum())], rtol=1e-12)
    assert m0[2] == adv.numel()


def test_eight_rank_dp_equals_single_process(tmp_path):
    """The world size the driver's scaling run ends at: eight gloo ranks, one env each (N = 8), every mini-batch the union of
    the eight ranks' rows -- identical parameters on all ranks, equal to the one-process update; the KL predicate, the moments
    reduction and `GradSync.any_` (the cross-rank agreement used around hipGraph captures) on an 8-rank group."""
    from oracle import ref_cpu as R
    name, world = "ppo_mlp_klskip", 8
    mp.spawn(_rank_main, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    assert c["N"] == world
    st = ppo_rollout(c, fx)
    st["returns"], st["advantages"] = t(fx["returns"]), t(fx["advantages"])
    keys = ("observations", "actions", "values", "returns", "actions_log_prob", "advantages", "mu", "sigma")
    p = state_dict_t(cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"]))
    ref = R.ppo_update(p, {k: st[k] for k in keys}, ppo_cfg(c), c["it"])
    ranks = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    assert all(np.array_equal(ranks[0], x) for x in ranks[1:]), "ranks diverged"
    assert_params_close(ranks[0], flat_state(p), c["lr"], len(ref["loss_trace"]))
    assert all(int(np.load(tmp_path / f"c{r}.npy")[0]) == ref["log"]["Train/kl_update_count"] for r in range(world))
    adv = st["advantages"].double().reshape(-1)
    for r in (0, world - 1):
        m = np.load(tmp_path / f"m{r}.npy")
        np.testing.assert_allclose(m[:2], [float(adv.sum()), float((adv * adv).sum())], rtol=1e-12)
        assert m[2] == adv.numel()
        assert list(np.load(tmp_path / f"a{r}.npy")) == [1.0, 0.0]          # any_(rank == 3) is true everywhere, any_(False) nowhere


def test_shard_envs_and_env_parsing(monkeypatch):
    from partmanip_amd import dist as pdist
    assert [pdist.shard_envs(32768, r, 8) for r in (0, 7)] == [(0, 4096), (28672, 32768)]
    with pytest.raises(ValueError):
        pdist.shard_envs(10, 0, 4)
    with pytest.raises(ValueError):
        pdist.shard_envs(4100, 0, 8)                           # 4096-env shards only from a multiple of the world size
    assert [pdist.shard_envs(4096, r, 8) for r in range(8)] == [(512 * r, 512 * (r + 1)) for r in range(8)]
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert pdist.init_from_env("gloo") == (0, 1, 0)            # single process: no group is created
    assert pdist.maybe_sync() is None


# ----------------------------------------------------------------------------------------------- DAgger under DP
# `dagger.update` has the same single all-reduce per optimiser step (partmanip_amd/algorithms/dagger.py: the flat
# student gradient with the loss in its tail).  Two ranks with env shards of the ring == one process with the whole
# ring and twice the mini-batch (sequential sampler; a mini-batch is a whole number of env steps on either side).
_DAG = dict(N=8, buf=4, O_s=24, O_t=16, A=6, n_minibatches=2, n_updates=2, lr=2e-3, seed=611,
            net=dict(name="MLP", hid_dim=[32, 32], activation="tanh"))


def _dagger_problem():
    from tests.golden.detgen import det_normal
    c = _DAG
    stu = state_dict_t(cases.actor_critic_state(c["net"], c["O_s"], c["A"], 0.1, c["seed"]))
    tea = state_dict_t(cases.actor_critic_state(c["net"], c["O_t"], c["A"], 0.5, c["seed"] + 1))
    obs = t(det_normal((c["buf"], c["N"], c["O_s"]), c["seed"] * 7 + 1))
    tobs = t(det_normal((c["buf"], c["N"], c["O_t"]), c["seed"] * 7 + 2))
    model = lambda std: dict(action_std=std, action_activate="tanh", clipAction=1.0, network=dict(c["net"]))
    cfg = dict(model=model(0.1), tea_model=model(0.5), n_updates=c["n_updates"], n_minibatches=c["n_minibatches"],
               sampler="sequential", lr=c["lr"], lr_schedule="fixed", max_iterations=100, proprio_shape=0)
    return stu, tea, obs, tobs, cfg


def _dagger_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import ref_cpu as R
    from partmanip_amd import dist as pdist
    pdist.init_from_env("gloo")
    sync = pdist.maybe_sync()
    stu, tea, obs, tobs, cfg = _dagger_problem()
    lo, hi = pdist.shard_envs(_DAG["N"], rank, world)
    ring = obs[:, lo:hi].reshape(-1, obs.shape[-1])              # this rank's ring: (buf * N/W, .) rows, step-major
    tring = tobs[:, lo:hi].reshape(-1, tobs.shape[-1])
    out = R.dagger_update(stu, tea, ring, tring, ring.shape[0], cfg, 1, grad_sync=OracleSyncAdapter(sync))
    np.save(os.path.join(out_dir, f"r{rank}.npy"), flat_state(stu))
    np.save(os.path.join(out_dir, f"l{rank}.npy"), np.array(out["loss_trace"]))
    sync.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_dagger_dp_equals_single_process(tmp_path):
    from oracle import ref_cpu as R
    mp.spawn(_dagger_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    stu, tea, obs, tobs, cfg = _dagger_problem()
    ring, tring = obs.reshape(-1, obs.shape[-1]), tobs.reshape(-1, tobs.shape[-1])
    ref = R.dagger_update(stu, tea, ring, tring, ring.shape[0], cfg, 1)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1), "ranks diverged"
    np.testing.assert_allclose(np.load(tmp_path / "l0.npy"), ref["loss_trace"], rtol=1e-5)
    assert_params_close(r0, flat_state(stu), _DAG["lr"], len(ref["loss_trace"]))


def test_resolve_seed_single_process():
    from partmanip_amd import dist as pdist
    assert pdist.resolve_seed(lambda: (7, "x_seed7")) == (7, "x_seed7")


def _seed_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from partmanip_amd import dist as pdist
    pdist.init_from_env("gloo")
    got = pdist.resolve_seed(lambda: (1000 + rank, f"run_seed{1000 + rank}"))     # every rank would pick its own
    sync = pdist.maybe_sync()
    w = torch.full((5,), float(rank + 1))
    sync.broadcast_(w)
    np.save(os.path.join(out_dir, f"s{rank}.npy"), np.array([got[0], float(w[0])]))
    torch.distributed.destroy_process_group()


def test_rank0_seed_and_parameters_reach_every_rank(tmp_path):
    """train.py: `seed: -1` must resolve to ONE seed / run name; the runners broadcast rank 0's parameters."""
    mp.spawn(_seed_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    s0, s1 = np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy")
    assert list(s0) == list(s1) == [1000.0, 1.0]


def _probe_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), PARTMANIP_PROBE_TIMEOUT="4")
    from partmanip_amd import dist as pdist
    pdist.init_from_env("gloo")
    a = pdist.maybe_sync(name="actor")
    c = pdist.GradSync(group=torch.distributed.new_group(), name="critic")
    c2 = pdist.GradSync(group=torch.distributed.new_group(), name="spare")
    a.mode = c.mode = c2.mode = "split"
    res = [a.probe("cpu")]                                         # healthy: None everywhere
    os.environ["PARTMANIP_TEST_COLLECTIVE_FAIL"] = "rank1:critic"
    res.append(c.probe("cpu"))                                     # rank 1 alone sees an error: EVERY rank must learn it
    os.environ["PARTMANIP_TEST_COLLECTIVE_FAIL"] = "rank1:spare:absent"
    res.append(c2.probe("cpu"))                                    # rank 1 never joins: rank 0 runs into the time box
    os.environ.pop("PARTMANIP_TEST_COLLECTIVE_FAIL")
    res.append(a.probe("cpu"))                                     # the first communicator still works (fresh store keys)
    x = torch.full((3,), float(rank + 1))
    a.sum_(x)
    # an all-reduce that fails past the probes names itself (here: a dtype no backend reduces)
    try:
        a._all_reduce(torch.zeros(2, dtype=torch.float8_e4m3fn))
        res.append("no error")
    except pdist.CollectiveError as e:
        res.append(str(e))
    except Exception as e:                                         # noqa: BLE001
        res.append("other: " + type(e).__name__)
    with open(os.path.join(out_dir, f"p{rank}.txt"), "w") as f:
        f.write(repr((res, x.tolist())))
    os._exit(0)                                                    # (rank 0's abandoned all-reduce of 'spare' never completes: no clean teardown)


def test_first_collective_of_every_communicator_is_probed_and_the_outcome_agreed(tmp_path):
    """dist.GradSync.probe: a failing first all-reduce on ONE rank is reported identically on EVERY rank
    (agreement runs over the rendezvous store, not over the communicator under test) with rank / communicator / launch structure /
    hint in the text; a rank that never joins shows up as a time-out on the others; a healthy communicator probes to None; a later
    failing all-reduce raises CollectiveError naming the call."""
    mp.spawn(_probe_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (eval(open(tmp_path / f"p{r}.txt").read()) for r in (0, 1))
    for (res, x), me in ((r0, 0), (r1, 1)):
        assert res[0] is None and res[3] is None and x == [3.0, 3.0, 3.0]
        assert "communicator 'critic' failed on 1 of 2 ranks" in res[1] and "rank 1: RuntimeError: PARTMANIP_TEST_COLLECTIVE_FAIL" in res[1]
        assert f"seen from rank {me} of 2" in res[1] and "launch structure split" in res[1] and "every rank reached the same collective" in res[1]
        assert "communicator 'spare' failed on 2 of 2 ranks" in res[2] and "rank 0: TimeoutError: no completion within 4 s" in res[2]
        assert res[4].startswith("all-reduce #") and "communicator 'actor'" in res[4] and f"rank {me} of 2" in res[4], res[4]
    assert r0[0][1].split(" -- seen from")[0] == r1[0][1].split(" -- seen from")[0]
