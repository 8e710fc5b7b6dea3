"""Data-parallel learner with the REAL HIP path: two ranks share cuda:0 (gloo moves the flat
gradient message; RCCL needs one device per rank, which a 1-GPU box cannot offer) and must
reproduce the single-process HIP run over the union of their env shards."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.golden import cases
from tests.helpers import load_fixture, ppo_cfg, ppo_rollout, t, flat_state, FakeEnv, FakeLogger, assert_flat_params_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run_hip(c, fx, lo, hi, out_path):
    from partmanip_amd.algorithms import ppo
    n = hi - lo
    cc = dict(c)
    cc["N"] = n
    with tempfile.TemporaryDirectory() as d:
        run = ppo(FakeEnv(n, {"normal_state": c["O"]}, c["A"]), ppo_cfg(cc, device=DEV), FakeLogger(d))
    sd = cases.actor_critic_state(c["net"], c["O"], c["A"], c["action_std"], c["seed"])
    run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    st = ppo_rollout(c, fx)
    for tt in range(c["T"]):
        s = lambda k: st[k][tt, lo:hi].to(DEV)
        run.storage.add_transitions(s("observations"), s("actions"), s("rewards")[:, 0], s("dones")[:, 0],
                                    s("succs")[:, 0], s("values"), s("actions_log_prob")[:, 0], s("mu"), s("sigma"))
    run.log_dict = {}
    run.curr_iter = c["it"]
    run.learn(st["last_values"][lo:hi].to(DEV))
    torch.cuda.synchronize()
    np.save(out_path, flat_state(run.actor_critic.state_dict()))
    return run


def _rank_main(rank, world, port, name, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from partmanip_amd import dist as pdist
    pdist.init_from_env("gloo")
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    lo, hi = pdist.shard_envs(c["N"], rank, world)
    run = _run_hip(c, fx, lo, hi, os.path.join(out_dir, f"r{rank}.npy"))
    assert run.sync is not None and run.sync.world == world
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name", ["ppo_mlp_default", "ppo_mlp_allon", "ppo_pn_maxmean"])
def test_two_ranks_on_one_gpu_match_single_process(name, tmp_path):
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    mp.spawn(_rank_main, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    _run_hip(c, fx, 0, c["N"], str(tmp_path / "single.npy"))
    r0, r1, single = (np.load(tmp_path / f) for f in ("r0.npy", "r1.npy", "single.npy"))
    assert np.array_equal(r0, r1), "ranks diverged"
    assert_flat_params_close("two ranks vs one process", r0, single, c["lr"], 16)


# ----------------------------------------------------------------------------------------------- small-step fast path under DP
_BIG = dict(N=128, T=64, O=53, A=10, lr=3e-4, n_mb=4, n_up=5, seed=941, net=dict(name="MLP", hid_dim=[256, 256], activation="tanh"))


def _big_problem():
    """A state-PPO rollout large enough for the small-step machinery to be live on every rank: 64 envs x 64 steps per rank ->
    1024-row mini-batches (split-K slabs > 1), O = 53 (16-byte padded observation rows), 5 epochs (hipGraph capture in epoch
    2, replay in 3-5).  Rank r's mini-batch k and the single process's mini-batch k cover the same 16 time steps."""
    from oracle import ref_cpu as R
    from tests.golden.detgen import det_normal
    from tests.test_gpu_fullsize import _cfg, _rollout_from_policy
    c = _BIG
    sd = cases.actor_critic_state(c["net"], c["O"], c["A"], 0.5, c["seed"])
    p = {k: t(v.copy()) for k, v in sd.items()}
    cfg = _cfg(c["net"], c["N"], c["T"], c["n_mb"], c["n_up"], c["lr"], "cpu")
    st = _rollout_from_policy(p, cfg["model"], t(det_normal((c["T"], c["N"], c["O"]), c["seed"] + 1)), c["seed"] + 2)
    return sd, st


def _run_big(lo, hi, out_path, expect_mode):
    from partmanip_amd.algorithms import ppo
    from tests.test_gpu_fullsize import _cfg
    c = _BIG
    sd, st = _big_problem()
    n = hi - lo
    with tempfile.TemporaryDirectory() as d:
        run = ppo(FakeEnv(n, {"normal_state": c["O"]}, c["A"]), _cfg(c["net"], n, c["T"], c["n_mb"], c["n_up"], c["lr"], DEV), FakeLogger(d))
    run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    assert run.use_graphs and run.solo_group and run.fused_head and run.dp_graph_mode == expect_mode, \
        (run.use_graphs, run.solo_group, run.fused_head, run.dp_graph_mode)
    for tt in range(c["T"]):
        s = lambda k: st[k][tt, lo:hi].to(DEV)
        run.storage.add_transitions(s("observations"), s("actions"), s("rewards")[:, 0], s("dones")[:, 0], s("succs")[:, 0], s("values"),
                                    s("actions_log_prob")[:, 0], s("mu"), s("sigma"))
    run.log_dict = {}
    run.curr_iter = 1
    run.learn(st["last_values"][lo:hi].to(DEV))
    torch.cuda.synchronize()
    n_graphs = sum(1 for k in run._graphs if isinstance(k, tuple))
    np.save(out_path, flat_state(run.actor_critic.state_dict()))
    return run, n_graphs


def _rank_big(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from partmanip_amd import dist as pdist
    pdist.init_from_env("gloo")
    lo, hi = pdist.shard_envs(_BIG["N"], rank, world)
    run, n_graphs = _run_big(lo, hi, os.path.join(out_dir, f"r{rank}.npy"), "split")
    assert run.sync is not None and run.sync.world == world and run.sync_c is not run.sync
    assert n_graphs == 4 * _BIG["n_mb"], n_graphs               # per mini-batch: {actor, critic} x {before, after the all-reduce}
    np.save(os.path.join(out_dir, f"log{rank}.npy"), np.array([float(run.log_dict[k]) for k in
                                                                ("Train/surrogate_loss", "Train/kl", "Train/value_function_loss", "Train/kl_update_count")]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_keep_the_small_step_fast_path(tmp_path):
    """with a process group the MLP learner used to drop hipGraph replay, the grouped weight-gradient /
    optimiser launches and the fused heads.  Now each rank keeps all three: the split-K slabs are folded before ONE all-reduce
    SUM per step, the optimiser's norm pass turns it into the mean and re-takes the KL predicate, and every step replays as
    [graph] -> all-reduce -> [graph] (gloo here; captured inside the graphs under RCCL).  Two ranks == one process."""
    c = _BIG
    mp.spawn(_rank_big, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    run, n_graphs = _run_big(0, c["N"], str(tmp_path / "single.npy"), None)
    assert n_graphs == 2                                           # one 4-step chunk per network
    r0, r1, single = (np.load(tmp_path / f) for f in ("r0.npy", "r1.npy", "single.npy"))
    assert np.array_equal(r0, r1), "ranks diverged"
    l0, l1 = np.load(tmp_path / "log0.npy"), np.load(tmp_path / "log1.npy")
    assert np.array_equal(l0, l1) and l0[3] == c["n_up"] * c["n_mb"] == run.log_dict["Train/kl_update_count"]
    want = [float(run.log_dict[k]) for k in ("Train/surrogate_loss", "Train/kl", "Train/value_function_loss")]
    np.testing.assert_allclose(l0[:3], want, rtol=2e-4, atol=2e-6)
    assert_flat_params_close("small-step fast path: two ranks vs one process", r0, single, c["lr"], c["n_up"] * c["n_mb"])


def _rank_big_rccl(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", PARTMANIP_FORCE_SYNC="1")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    run, n_graphs = _run_big(0, _BIG["N"], os.path.join(out_dir, "rccl.npy"), "capture")
    assert n_graphs == 2, n_graphs                                  # the all-reduces are INSIDE the two 4-step graphs
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_rccl_all_reduce_captured_inside_the_step_graphs(tmp_path):
    """Backend nccl (= RCCL): the gradient all-reduces are captured into the multi-step hipGraphs (one RCCL rank is all a
    1-GPU box can host; the collectives still go through librccl and through graph capture / replay)."""
    mp.spawn(_rank_big_rccl, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    _run_big(0, _BIG["N"], str(tmp_path / "single.npy"), None)
    got, single = np.load(tmp_path / "rccl.npy"), np.load(tmp_path / "single.npy")
    assert np.array_equal(got, single), "a one-rank all-reduce SUM with scale 1.0 must not change a bit"


# ----------------------------------------------------------------------------------------------- overlap=True under DP
def _rank_overlap(rank, world, port, name, out_dir):
    os.environ["PARTMANIP_OVERLAP"] = "1"          # actor and critic steps (and their all-reduces) on two HIP streams
    _rank_main(rank, world, port, name, out_dir)


def test_two_ranks_with_actor_critic_overlap_match_single_process(tmp_path):
    """The fused point-cloud backbones default to the serial order; with PARTMANIP_OVERLAP=1 the critic step (and its
    all-reduce) is issued on a second stream next to the actor's.  Same result as one process, ranks identical."""
    name = "ppo_pn_maxmean"
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    mp.spawn(_rank_overlap, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    _run_hip(c, fx, 0, c["N"], str(tmp_path / "single.npy"))
    r0, r1, single = (np.load(tmp_path / f) for f in ("r0.npy", "r1.npy", "single.npy"))
    assert np.array_equal(r0, r1), "ranks diverged"
    assert_flat_params_close("two ranks vs one process", r0, single, c["lr"], 16)


# ----------------------------------------------------------------------------------------------- RCCL itself, one rank
def _rank_rccl(rank, world, port, name, out_dir):
    """A 1-GPU box cannot host two RCCL ranks, but it can host ONE: backend "nccl" (= RCCL on ROCm), world size 1,
    PARTMANIP_FORCE_SYNC=1 -> every all-reduce / broadcast of the data-parallel learner goes through librccl."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      PARTMANIP_FORCE_SYNC="1")
    from partmanip_amd import dist as pdist
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert torch.distributed.get_backend() == "nccl"
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    run = _run_hip(c, fx, 0, c["N"], os.path.join(out_dir, "rccl.npy"))
    assert isinstance(run.sync, pdist.GradSync) and run.sync.world == 1
    probe = torch.arange(8, dtype=torch.float32, device=DEV)
    run.sync.mean_(probe)
    assert torch.equal(probe.cpu(), torch.arange(8, dtype=torch.float32))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("name", ["ppo_mlp_allon", "ppo_pn_maxmean"])
def test_learner_collectives_run_on_rccl(name, tmp_path):
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    mp.spawn(_rank_rccl, args=(1, _free_port(), name, str(tmp_path)), nprocs=1, join=True)
    _run_hip(c, fx, 0, c["N"], str(tmp_path / "single.npy"))
    got, single = np.load(tmp_path / "rccl.npy"), np.load(tmp_path / "single.npy")
    # the synchronised path recomputes the KL flag after the reduce and runs without stream overlap: same arithmetic per
    # kernel, so only launch-order-independent differences (none expected) remain
    assert_flat_params_close("RCCL world 1 vs no process group", got, single, c["lr"], 16)


# ----------------------------------------------------------------------------------------------- DAgger under DP (HIP)
_DAG = dict(N=8, buf=4, O_s=24, O_t=16, A=6, n_minibatches=2, n_updates=2, lr=2e-3, seed=611,
            net=dict(name="MLP", hid_dim=[32, 32], activation="tanh"))


def _dagger_cfg(n_envs, teacher):
    c = _DAG
    return dict(num_envs=n_envs, obs_mode="stu_mode",
                model=dict(action_std=0.1, action_activate="tanh", clipAction=1.0, network=dict(c["net"])),
                max_iterations=100, n_steps=1, n_updates=c["n_updates"], n_minibatches=c["n_minibatches"], device=DEV,
                buf_size=c["buf"], reward_reset=False, add_proprio_obs=False, offline_data_pth=None, eval_round=1,
                eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
                lr_schedule="fixed", lr=c["lr"], teacher=teacher, resume=None, pretrain=None, sampler="sequential")


def _write_teacher(d):
    from partmanip_amd.algorithms import ppo
    c = _DAG
    tc = dict(net=c["net"], N=c["N"], T=1, n_updates=1, n_minibatches=1, tricks=dict(cases.TRICKS_DEFAULT),
              sampler="sequential", succ_value=None, lr=1e-3, desired_kl=0.1, lr_schedule="fixed", gamma=0.99, lam=0.95,
              epsilon_clip=0.2, action_std=0.5, max_iterations=10, O=c["O_t"], A=c["A"])
    tea = ppo(FakeEnv(c["N"], {"normal_state": c["O_t"]}, c["A"]), ppo_cfg(tc, device=DEV), FakeLogger(d))
    tea.actor_critic.load_state_dict({k: t(v.copy()) for k, v in
                                      cases.actor_critic_state(c["net"], c["O_t"], c["A"], 0.5, c["seed"] + 1).items()})
    tea.save(1)
    return os.path.join(d, "model_1.pth")


def _run_dagger_hip(lo, hi, teacher, out_path, d):
    from partmanip_amd.algorithms import dagger
    from tests.golden.detgen import det_normal
    c = _DAG
    n = hi - lo
    env = FakeEnv(n, {"stu_mode": c["O_s"], "normal_state": c["O_t"], "proprio_state": 0}, c["A"])
    sd = {k: t(v.copy()) for k, v in cases.actor_critic_state(c["net"], c["O_s"], c["A"], 0.1, c["seed"]).items()}
    run = dagger(env, _dagger_cfg(n, teacher), FakeLogger(d))
    run.student.load_state_dict(sd)
    obs = t(det_normal((c["buf"], c["N"], c["O_s"]), c["seed"] * 7 + 1))
    tobs = t(det_normal((c["buf"], c["N"], c["O_t"]), c["seed"] * 7 + 2))
    for k in range(c["buf"]):
        run.storage.add_transitions_dagger(obs[k, lo:hi].to(DEV), tobs[k, lo:hi].to(DEV))
    run.log_dict = {}
    run.update(1)
    torch.cuda.synchronize()
    np.save(out_path, flat_state(run.student.state_dict()))
    return run


def _rank_dagger(rank, world, port, teacher, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from partmanip_amd import dist as pdist
    pdist.init_from_env("gloo")
    lo, hi = pdist.shard_envs(_DAG["N"], rank, world)
    run = _run_dagger_hip(lo, hi, teacher, os.path.join(out_dir, f"r{rank}.npy"), out_dir)
    assert run.sync is not None and run.sync.world == world
    np.save(os.path.join(out_dir, f"l{rank}.npy"), np.array([run.log_dict["Train/dagger_loss"]]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_dagger_update_matches_single_process(tmp_path):
    """`dagger.update`'s gradient all-reduce (dagger.py `sync.mean_`): two ranks with env shards of the ring == one
    process with the whole ring and twice the mini-batch; the reported loss is the global mean."""
    teacher = _write_teacher(str(tmp_path))
    mp.spawn(_rank_dagger, args=(2, _free_port(), teacher, str(tmp_path)), nprocs=2, join=True)
    run = _run_dagger_hip(0, _DAG["N"], teacher, str(tmp_path / "single.npy"), str(tmp_path))
    r0, r1, single = (np.load(tmp_path / f) for f in ("r0.npy", "r1.npy", "single.npy"))
    assert np.array_equal(r0, r1), "ranks diverged"
    np.testing.assert_allclose(np.load(tmp_path / "l0.npy")[0], run.log_dict["Train/dagger_loss"], rtol=2e-6)
    np.testing.assert_allclose(np.load(tmp_path / "l1.npy")[0], run.log_dict["Train/dagger_loss"], rtol=2e-6)
    assert_flat_params_close("dagger: two ranks vs one process", r0, single, _DAG["lr"], 4)


# ----------------------------------------------------------------------------------------------- train.py under DP
def _rank_train(rank, world, port, log_root, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), PARTMANIP_SHARE_GPU="1", PARTMANIP_DIST_BACKEND="gloo")
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.argv = ["train.py", "--algocfg", "ppo", "--taskcfg", "open_drawer", "--exp_name", "dp", "--algo.num_envs", "32",
                "--algo.n_steps", "4", "--algo.max_iterations", "2", "--algo.n_minibatches", "2", "--algo.save_frequence", "2",
                "--log.log_root", log_root]
    np.random.seed(100 + rank)                                   # `seed: -1` (base_cfg.yaml) draws from numpy: ranks would differ
    import train
    runner = train.main()
    assert runner.sync is not None and runner.vec_env.num_envs == 16
    np.save(os.path.join(out_dir, f"w{rank}.npy"), flat_state(runner.actor_critic.state_dict()))
    rms = runner.state_norm.running_ms
    np.save(os.path.join(out_dir, f"rms{rank}.npy"), torch.cat([rms.mean, rms.std], 0).cpu().numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_train_py_two_ranks_random_seed_keeps_one_model(tmp_path):
    """with the default `seed: -1` every rank used to draw its own seed -> different initial weights,
    run names and checkpoint directories.  Now rank 0 resolves the seed, its parameters / Adam state / observation
    statistics are broadcast, the running mean/std are updated from all-reduced moments, and rank 0 alone saves."""
    log_root = str(tmp_path / "logs")
    mp.spawn(_rank_train, args=(2, _free_port(), log_root, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    assert np.array_equal(w0, w1), "replicas hold different models"
    assert np.array_equal(np.load(tmp_path / "rms0.npy"), np.load(tmp_path / "rms1.npy")), "observation statistics differ"
    ck = [os.path.join(d, f) for d, _, fs in os.walk(log_root) for f in fs if f.endswith(".pth")]
    assert len(ck) == 1 and ck[0].endswith("model_2.pth"), ck           # one run directory, written once


# ----------------------------------------------------------------------------------------------- bench.py --gpus N
def test_bench_self_launches_n_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself and prints ONE line with
    n_gpus = 2 (here both ranks share the box's single GPU over gloo); without that override it refuses (exit 2)
    because fewer than 2 devices are visible -- it must never silently measure one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "state",
           "--n-steps", "16", "--no-cpu-baseline"]
    if torch.cuda.device_count() < 2:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 2 and "GPU(s) are visible" in out.stderr, (out.returncode, out.stderr[-500:])
    env.update(PARTMANIP_SHARE_GPU="1", PARTMANIP_DIST_BACKEND="gloo")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["world_size_observed"] == 2 and rec["config"]["parallelism"] == "dp2"
    assert rec["value"] > 0
    dp = rec["config"]["data_parallel"]                           # a multi-GPU line explains itself
    assert dp["param_checksum_equal"] is True and len(dp["per_rank_ms_per_step"]) == 2 and dp["all_reduces_per_step"] > 0
    assert rec["comm_ms_per_step"] == dp["comm_ms_per_step"] >= 0.0 and dp["graph_mode"] in ("split", "capture")
    assert rec["config"]["backend"] == "gloo"


def test_bench_degrades_when_a_communicator_fails_on_one_rank(tmp_path):
    """the first multi-device run must be diagnosable.  The first all-reduce of every communicator is
    time-boxed and its outcome agreed over the rendezvous store (dist.GradSync.probe); here rank 1's probe of the SECOND
    communicator ('critic') is made to fail: both ranks must print the diagnosis (rank, device, communicator, launch structure,
    the NCCL_DEBUG / env hints), fall back TOGETHER to one communicator + one stream + eager launches, finish with equal
    parameters, and the line must say so in config.data_parallel.degraded.  A failure of the DEFAULT communicator has nothing to
    fall back to: the run stops with the diagnosis and prints no line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PARTMANIP_SHARE_GPU="1", PARTMANIP_DIST_BACKEND="gloo", PARTMANIP_TEST_COLLECTIVE_FAIL="rank1:critic", PARTMANIP_PROBE_TIMEOUT="20")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "state",
           "--n-steps", "16", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    dp = json.loads(lines[0])["config"]["data_parallel"]
    assert dp["degraded"] and "communicator 'critic'" in dp["degraded"]["reason"] and "rank 1: RuntimeError" in dp["degraded"]["reason"]
    assert "one communicator" in dp["degraded"]["mode"] and dp["graph_mode"].startswith("eager (degraded")
    assert dp["param_checksum_equal"] is True and dp["all_reduces_per_step"] > 0
    assert out.stderr.count("DEGRADED: retrying with ONE communicator") == 2          # every rank said so
    assert "NCCL_DEBUG" in out.stderr or "every rank reached the same collective" in out.stderr
    # the default communicator: nothing to degrade to
    env["PARTMANIP_TEST_COLLECTIVE_FAIL"] = "rank1:actor"
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "data-parallel PPO cannot start" in out.stderr and "communicator 'actor'" in out.stderr and "rank 1: RuntimeError" in out.stderr


def test_bench_eight_ranks_on_the_headline_workload(tmp_path):
    """The driver's scaling run ends at `bench.py --gpus 8` on the headline (vision) workload, and nothing had ever started eight
    ranks of it.  Here all eight share the box's one GPU over gloo (T = 1 so that it takes a minute): eight processes rendezvous,
    shard nothing (weak scaling: 4096 envs each), exchange one all-reduce per optimiser step on two communicators, end with equal
    parameters, and rank 0 prints ONE line that names every rank's device."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PARTMANIP_SHARE_GPU="1", PARTMANIP_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--workload", "vision",
           "--n-steps", "1", "--no-cpu-baseline", "--no-optional", "--lean"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert rec["n_gpus"] == 8 and cfg["world_size_observed"] == 8 and cfg["parallelism"] == "dp8" and cfg["backend"] == "gloo"
    assert rec["scaling"] == "weak" and rec["value"] > 0
    dp = cfg["data_parallel"]
    assert dp["param_checksum_equal"] is True and len(dp["per_rank_ms_per_step"]) == 8
    # T = 1: 4096 rows per rank, n_minibatches 8 -> 512-row mini-batches: 8 per epoch x 5 epochs x {actor, critic}
    assert dp["all_reduces_per_step"] == 80, dp["all_reduces_per_step"]
    ranks = cfg["ranks"]
    assert ranks["world"] == 8 and [r["rank"] for r in ranks["ranks"]] == list(range(8)) and len({r["pid"] for r in ranks["ranks"]}) == 8
    assert "[bench] 8 ranks over gloo" in out.stderr and "rank 7: pid" in out.stderr


# ----------------------------------------------------------------------------------------------- two REAL RCCL ranks (>= 2 GPUs)
# A 1-GPU box skips these; on a multi-GPU node (the driver's SCALE run has one) they are the first place where RCCL carries the
# learner's collectives between two devices: PPO-MLP with the hipGraph fast path (all-reduce captured inside the graphs),
# PPO-PointNet and DAgger, each equal to the single-process run over the union of the env shards.
needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="two RCCL ranks need two GPUs")


def _rccl_env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    os.environ.pop("PARTMANIP_DIST_BACKEND", None)
    os.environ.pop("PARTMANIP_SHARE_GPU", None)
    from partmanip_amd import dist as pdist
    r, w, local = pdist.init_from_env("nccl")
    assert torch.distributed.get_backend() == "nccl" and w == world and local == rank
    torch.cuda.set_device(local)
    return pdist


def _rank_main_rccl(rank, world, port, name, out_dir):
    global DEV
    pdist = _rccl_env(rank, world, port)
    DEV = f"cuda:{rank}"
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    lo, hi = pdist.shard_envs(c["N"], rank, world)
    run = _run_hip(c, fx, lo, hi, os.path.join(out_dir, f"r{rank}.npy"))
    assert run.sync is not None and run.sync.world == world
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@needs_two_gpus
@pytest.mark.parametrize("name", ["ppo_mlp_allon", "ppo_pn_maxmean"])
def test_two_rccl_ranks_match_single_process(name, tmp_path):
    c, fx = cases.PPO_CASES[name], load_fixture(name)
    mp.spawn(_rank_main_rccl, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    _run_hip(c, fx, 0, c["N"], str(tmp_path / "single.npy"))
    r0, r1, single = (np.load(tmp_path / f) for f in ("r0.npy", "r1.npy", "single.npy"))
    assert np.array_equal(r0, r1), "ranks diverged"
    assert_flat_params_close("two RCCL ranks vs one process", r0, single, c["lr"], 16)


def _rank_big_rccl2(rank, world, port, out_dir):
    global DEV
    pdist = _rccl_env(rank, world, port)
    DEV = f"cuda:{rank}"
    lo, hi = pdist.shard_envs(_BIG["N"], rank, world)
    run, n_graphs = _run_big(lo, hi, os.path.join(out_dir, f"r{rank}.npy"), "capture")
    assert run.sync.world == world and run.sync_c is not run.sync
    assert n_graphs == 2 and run.dp_graph_mode == "capture", (n_graphs, run.dp_graph_mode)   # all-reduces INSIDE the two 4-step graphs
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@needs_two_gpus
def test_two_rccl_ranks_keep_the_graph_fast_path(tmp_path):
    """The default data-parallel mode of the small-step regime under RCCL -- gradient all-reduces captured INSIDE the multi-step
    hipGraphs, the critic's collectives on their own communicator -- with a second rank on a second GPU."""
    c = _BIG
    mp.spawn(_rank_big_rccl2, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    _run_big(0, c["N"], str(tmp_path / "single.npy"), None)
    r0, r1, single = (np.load(tmp_path / f) for f in ("r0.npy", "r1.npy", "single.npy"))
    assert np.array_equal(r0, r1), "ranks diverged"
    assert_flat_params_close("graph fast path: two RCCL ranks vs one process", r0, single, c["lr"], c["n_up"] * c["n_mb"])


def _rank_dagger_rccl(rank, world, port, teacher, out_dir):
    global DEV
    pdist = _rccl_env(rank, world, port)
    DEV = f"cuda:{rank}"
    lo, hi = pdist.shard_envs(_DAG["N"], rank, world)
    run = _run_dagger_hip(lo, hi, teacher, os.path.join(out_dir, f"r{rank}.npy"), out_dir)
    assert run.sync is not None and run.sync.world == world
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@needs_two_gpus
def test_two_rccl_rank_dagger_update_matches_single_process(tmp_path):
    teacher = _write_teacher(str(tmp_path))
    mp.spawn(_rank_dagger_rccl, args=(2, _free_port(), teacher, str(tmp_path)), nprocs=2, join=True)
    _run_dagger_hip(0, _DAG["N"], teacher, str(tmp_path / "single.npy"), str(tmp_path))
    r0, r1, single = (np.load(tmp_path / f) for f in ("r0.npy", "r1.npy", "single.npy"))
    assert np.array_equal(r0, r1), "ranks diverged"
    assert_flat_params_close("dagger: two RCCL ranks vs one process", r0, single, _DAG["lr"], 4)


@needs_two_gpus
def test_bench_two_rccl_ranks_report_the_data_parallel_block():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PARTMANIP_SHARE_GPU", "PARTMANIP_DIST_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                          "--no-optional"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    dp = rec["config"]["data_parallel"]
    assert rec["n_gpus"] == 2 and rec["config"]["backend"] == "nccl" and dp["param_checksum_equal"] and dp["comm_ms_per_step"] > 0


# ----------------------------------------------------------------------------------------------- capture failure -> eager
def _rank_capture_fails(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", PARTMANIP_FORCE_SYNC="1",
                      PARTMANIP_TEST_CAPTURE_FAIL="1")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    run, n_graphs = _run_big(0, _BIG["N"], os.path.join(out_dir, "fallback.npy"), "capture")
    # (the launch-structure name stays what it was -- the '== "split"' checks of the update depend on it; the text is separate)
    assert n_graphs == 0 and str(run.graph_status).startswith("eager (capture") and run.dp_graph_mode == "capture", \
        (n_graphs, run.graph_status, run.dp_graph_mode)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_a_failing_graph_capture_falls_back_to_eager_steps(tmp_path):
    """If capturing a step (with its RCCL all-reduce) into a hipGraph throws -- forced here -- the learner must neither die nor
    skip the step: the step runs eagerly, capturing stops, `graph_status` says what happened (bench.py prints it in the
    line's data_parallel block), and the update is the one the graphs would have produced."""
    mp.spawn(_rank_capture_fails, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    _run_big(0, _BIG["N"], str(tmp_path / "single.npy"), None)
    got, single = np.load(tmp_path / "fallback.npy"), np.load(tmp_path / "single.npy")
    assert np.array_equal(got, single)


def _rank_one_capture_fails(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      PARTMANIP_TEST_CAPTURE_FAIL="rank1")
    from partmanip_amd import dist as pdist
    pdist.init_from_env("gloo")
    lo, hi = pdist.shard_envs(_BIG["N"], rank, world)
    run, n_graphs = _run_big(lo, hi, os.path.join(out_dir, f"f{rank}.npy"), "split")
    # rank 1's first capture throws; rank 0's succeeds -- and is dropped: both continue eagerly, neither keeps a graph
    assert n_graphs == 0 and str(run.graph_status).startswith("eager (split: capture failed"), (rank, n_graphs, run.graph_status)
    assert ("another rank" in run.graph_status) == (rank == 0), (rank, run.graph_status)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_a_capture_that_fails_on_one_rank_only_moves_every_rank_to_eager_steps(tmp_path):
    """a capture can fail on ONE rank (memory); ranks that then took different paths would issue the collectives of a
    communicator in different orders.  The ranks agree on the outcome of every capture (GradSync.any_) before choosing: here rank
    1's capture is made to throw, rank 0 discards its good graph, and the two-rank update still equals the one-process update."""
    c = _BIG
    mp.spawn(_rank_one_capture_fails, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    _run_big(0, c["N"], str(tmp_path / "single.npy"), None)
    f0, f1, single = (np.load(tmp_path / f) for f in ("f0.npy", "f1.npy", "single.npy"))
    assert np.array_equal(f0, f1), "ranks diverged"
    assert_flat_params_close("one-rank capture failure: two ranks vs one process", f0, single, c["lr"], c["n_up"] * c["n_mb"])
