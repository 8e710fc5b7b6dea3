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


def test_shard_envs_and_env_parsing(monkeypatch):
    from partmanip_amd import dist as pdist
    assert [pdist.shard_envs(32768, r, 8) for r in (0, 7)] == [(0, 4096), (28672, 32768)]
    with pytest.raises(ValueError):
        pdist.shard_envs(10, 0, 4)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert pdist.init_from_env("gloo") == (0, 1, 0)            # single process: no group is created
    assert pdist.maybe_sync() is None
