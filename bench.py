#!/usr/bin/env python3
"""bench.py -- PPO learner throughput on MI355X (BASELINE.json metric: PPO env-steps/sec,
4096 envs x 1024-pt clouds).

    python bench.py [--gpus N --steps K --warmup W] [--workload vision|state]

A "step" is one learner iteration = the reference's `learn_time` window (ppo.py:256-262:
compute_returns + update + storage.clear) over one synthetic rollout batch that is already
resident in HBM.  env-steps/s = N_env * T * world / (time per step).  One process per GPU;
for N > 1 launch under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE from the env):
each rank owns its own 4096 envs (weak scaling) and gradients are all-reduced over RCCL once
per optimiser step.

Rank 0 prints ONE JSON line including
  roofline     : live HIP-event timing of the dominant kernel (fused PointNet encoder forward)
                 against the fp32-MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
  cpu_baseline : the CPU oracle (oracle/ref_cpu.py, pinned to the reference) timed on this
                 box's host cores on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3
ENC_MAC_PER_POINT = 3 * 128 + 128 * 256 + 256 * 512          # 164 224

WORKLOADS = {
    # cfg3: open_drawer vision PPO, ppo.yaml hyper-parameters, PointNet(tanh, max_mean, no sub_mean)
    "vision": dict(name="ppo_vision_pointnet_4096env_x_8step_x_1024pt", N=4096, T=8, O=3072, A=10,
                   net=dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False)),
    # same rollouts through the PointNet++ (SSG) plug-in backbone: FPS + ball query + grouped shared MLPs
    "vision_pn2": dict(name="ppo_vision_pointnet2ssg_4096env_x_8step_x_1024pt", N=4096, T=8, O=3072, A=10,
                       net=dict(name="PointNet2", activation="tanh")),
    # cfg2: open_drawer state PPO, MLP 53-512-512-512
    "state": dict(name="ppo_state_mlp_4096env_x_128step", N=4096, T=128, O=53, A=10,
                  net=dict(name="MLP", hid_dim=[512, 512, 512], activation="tanh")),
}


DAGGER = dict(name="dagger_pointnet_student_4096env_x_16buf_x_{P}pt", N=4096, buf=16, O_t=53, A=10)


def run_dagger(args, device, rank, world):
    """cfg 5 analogue: DAgger, N=4096, n_steps 1, buf_size 16, PointNet student on 4096-pt
    clouds (`--points`; the reference hard-codes 1024, network.py:146; the 3D-Sparse-UNet cfg 5 names does not exist
    in the reference), frozen MLP teacher (O=53), random sampler, n_updates 2, n_minibatches 16 (-> 2048).
    A step = one `dagger.update` over the full ring (65 536 rows); env-steps/s = N * n_steps / time."""
    import tempfile
    from partmanip_amd.algorithms import ppo, dagger
    from partmanip_amd.feeder import FeederEnv, ScreenLogger
    conv = args.student == "conv3d"
    if conv:
        # the reference's SHIPPED DAgger configuration (cfg/algos/dagger_tsdf.yaml): 16 envs, 1600-step ring (25 600 rows of
        # a 50^3 TSDF + proprio = 12.8 GB, resident in HBM), Conv3DNet student, 16 mini-batches (-> 1600 rows), 2 passes
        d = dict(DAGGER, N=16, buf=1600, O_s=50 ** 3 + 25, proprio=25, name="dagger_conv3dnet_student_16env_x_1600buf_x_50cube_tsdf")
        obs_mode = "mesh_tsdf"
    else:
        d = dict(DAGGER, O_s=3 * args.points, proprio=0, name=DAGGER["name"].format(P=args.points))
        obs_mode = "depth_pc"
    torch.manual_seed(1234)
    tmp = tempfile.mkdtemp()
    env = FeederEnv(d["N"], {"normal_state": d["O_t"], obs_mode: d["O_s"], "proprio_state": d["proprio"]}, d["A"], device,
                    seed=1234 + rank, point_num=args.points)
    tcfg = make_cfg(WORKLOADS["state"], device)
    tcfg.update(num_envs=d["N"], n_steps=1, obs_mode="normal_state")
    tea = ppo(env, tcfg, ScreenLogger(tmp, "tea", "n", quiet=True))
    tea.save(1)
    net = (dict(name="Conv3DNet", activation="tanh") if conv else
           dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False, point_num=args.points))
    cfg = dict(num_envs=d["N"], obs_mode=obs_mode,
               model=dict(action_std=0.1, action_activate="tanh", clipAction=1.0, network=net),
               max_iterations=10000, n_steps=1, n_updates=2, n_minibatches=16, device=device, buf_size=d["buf"],
               reward_reset=False, add_proprio_obs=conv, offline_data_pth=None, eval_round=1, eval_frequence=10 ** 9,
               save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False, lr_schedule="fixed", lr=5e-5,
               teacher=os.path.join(tea.save_ckpt_dir, "model_1.pth"), resume=None, pretrain=None, sampler="random")
    run = dagger(env, cfg, ScreenLogger(tmp, "stu", "n", quiet=True))
    for _ in range(d["buf"]):
        obs = env.reset()
        run.storage.add_transitions_dagger(obs[obs_mode], obs["normal_state"])      # rows = [observation | proprio]
    run.log_dict = {}

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        run.update(1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.update(1)
    fence()
    dt = time.perf_counter() - t0
    rows = d["N"] * d["buf"] * 2                        # samples through the student per update
    return dict(metric="DAgger update throughput (shipped dagger_tsdf.yaml)" if conv else "DAgger update throughput (cfg 5 analogue)",
                value=d["N"] * 1 * world / (dt / args.steps),
                unit="env-steps/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32",
                data="synthetic", config=dict(workload=d["name"], ring_rows=d["N"] * d["buf"],
                                              minibatch=min(d["N"] * d["buf"] // 16, 2048),
                                              student_samples_per_s=rows * world / (dt / args.steps),
                                              dagger_loss=float(run.log_dict["Train/dagger_loss"])))


def make_cfg(w, device):
    return dict(num_envs=w["N"], obs_mode="obs", succ_value=None,
                model=dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=dict(w["net"])),
                max_iterations=200000, n_steps=w["T"], n_updates=5, n_minibatches=8, device=device, eval_round=1,
                eval_frequence=10 ** 9, save_frequence=10 ** 9, test_only=False, save_pose=False, save_video=False,
                lr_schedule="fixed", lr=5e-5, desired_kl=0.1, epsilon_clip=0.2, gamma=0.99, lam=0.95,
                tricks=dict(mini_adv_norm=False, whole_adv_norm=False, use_state_norm=False,
                            use_clipped_value_loss=False, use_grad_clip=True, max_grad_norm=0.5),
                sampler="sequential", resume=None)


def cpu_baseline(w, rollout_cpu, sd_cpu, cfg):
    """Time the oracle's actor + critic mini-batch passes on a bounded sample and extrapolate to a
    full iteration: per-sample cost x (n_updates * T*N samples) for each of the two loops."""
    from oracle import ref_cpu as R
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    vision = w["net"]["name"] in ("PointNet", "PointNet2")
    mb = 128 if vision else 2048                     # CPU PointNet pass: ~6 s per 128 samples
    n_mb = 1 if vision else 4
    T, N = w["T"], w["N"]
    keys = ("observations", "actions", "values", "returns", "actions_log_prob", "advantages", "mu", "sigma")
    sub = {k: rollout_cpu[k].reshape(-1, rollout_cpu[k].shape[-1])[: mb * n_mb].reshape(n_mb, mb, -1).clone() for k in keys}
    p = {k: v.clone() for k, v in sd_cpu.items()}
    c = dict(cfg)
    c.update(n_updates=1, n_minibatches=n_mb, sampler="sequential", device="cpu")
    t0 = time.perf_counter()
    R.gae_returns(rollout_cpu["rewards_full"], rollout_cpu["values_full"], rollout_cpu["dones_full"],
                  rollout_cpu["succs_full"], rollout_cpu["last_values"], 0.99, 0.95, None, False)
    t_gae = time.perf_counter() - t0
    st = {k: sub[k] for k in keys}
    t0 = time.perf_counter()
    R.ppo_update(p, st, c, 1)                         # n_mb actor steps + n_mb critic steps
    t_upd = time.perf_counter() - t0
    per_sample = t_upd / (mb * n_mb)                  # one actor pass + one critic pass of one sample
    t_iter = t_gae + per_sample * cfg["n_updates"] * T * N
    return dict(value=T * N / t_iter, unit="env-steps/s", cores=ncores, kind="port",
                sample=f"oracle/ref_cpu.py ppo_update on {n_mb} actor + {n_mb} critic mini-batches of {mb} samples "
                       f"({t_upd:.1f} s) + full GAE ({t_gae * 1e3:.1f} ms), extrapolated to "
                       f"{cfg['n_updates']} epochs x {T * N} samples; {ncores} torch threads")


def run_depth2pc(args, device):
    """Observation-side step with the only timing the reference publishes: the cloud sampling
    of `TSDFVolume.depth2pc` for 64 envs x 6 views x 180 x 320 px -> 1024 points, "slow.. ~0.5s"
    (utils/depth2tsdf.py:158, unstated NVIDIA GPU).  A step = one depth2pc call (back-projection, crop, compaction,
    FPS, gather); `sampling_ms` isolates the part the reference's comment is about."""
    from partmanip_amd import ops
    from partmanip_amd.depth2tsdf import TSDFVolume
    b, m, h, w_ = 64, 6, 180, 320
    vol = TSDFVolume(device)
    pose = torch.eye(4).repeat(m, 1, 1)
    for i in range(m):
        pose[i, :3, 3] = torch.tensor([0.02 * i, -0.01 * i, -0.6])
    vol.register_camera(pose.numpy(), [[250.0, 0.0, w_ / 2 - 0.5], [0.0, 250.0, h / 2 - 0.5], [0.0, 0.0, 1.0]], h, w_, b)
    torch.manual_seed(0)
    depth = torch.rand(b, m, h, w_, device=device) * 0.5 + 0.45

    def timed(fn, n):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    dt = timed(lambda: vol.depth2pc(depth), args.steps)
    lo = vol._vol_origin.cpu().numpy()
    world = ops.depth_backproject(depth, vol.cam_pose, 250.0, 250.0, w_ / 2 - 0.5, h / 2 - 0.5, lo, vol._size + lo)
    ws = ops.Workspace(device)

    def sample():
        c, n = ops.depth_compact(world)
        ops.group_points(c, ops.fps_varlen(c, n, 1024, ws).view(b, 1024, 1))
    ds = timed(sample, args.steps)
    valid = float((world != 0).any(-1).float().mean())
    # the other sampling call site (depth2tsdf.py:88-120, the 'depth_sparse' observation) on a scene with a surface:
    # a tilted plane seen by all views, 50^3 grid -> integrate + band select + FPS(1024) + gather
    yy, xx = torch.meshgrid(torch.arange(h, device=device), torch.arange(w_, device=device), indexing="ij")
    plane = (0.75 + 0.0004 * xx + 0.0006 * yy).float().expand(b, m, h, w_).contiguous()
    band = vol.integrate(plane).abs().lt(0.2).flatten(1).sum(-1).float().mean().item()
    dv = timed(lambda: vol.sparse_voxel(plane), args.steps)
    return dict(metric="depth2pc seconds per call, 64 envs x 6 views x 180x320 px -> 1024 pts", value=dt, unit="s",
                n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=dt * 1e3, higher_is_better=False,
                scaling="weak", vs_baseline=ds / 0.5, dtype="f32", data="synthetic",
                config=dict(workload="depth2pc_64env_x_6view_x_180x320", sampling_ms=ds * 1e3, in_crop_fraction=valid,
                            sparse_voxel_ms=dv * 1e3, sparse_voxel_band_voxels=band,
                            baseline="'~0.5s' for the sampling alone, utils/depth2tsdf.py:158 (BASELINE.md); "
                                     "vs_baseline = sampling time / 0.5 s"))


def build_runner(w, cfg, device, rank):
    """Runner + one synthetic rollout produced by the freshly initialised policy (ratio ~ 1, KL ~ 0) + the
    timed step: restore the initial policy / optimiser state, then the reference's `learn` window."""
    from partmanip_amd.algorithms import ppo
    from partmanip_amd.feeder import FeederEnv, ScreenLogger
    torch.manual_seed(1234)                            # identical initial weights on every rank
    env = FeederEnv(w["N"], {"obs": w["O"]}, w["A"], device, seed=1234 + rank)
    run = ppo(env, cfg, ScreenLogger(quiet=True))
    ac, st = run.actor_critic, run.storage
    obs = env.reset()["obs"]
    for _ in range(w["T"]):
        actions, logp, values, mu, sigma = ac.random_act_cri(obs)
        nxt, rew, done, _ = env.step(actions)
        st.add_transitions(obs, actions, rew, done, env.reset_succ, values, logp, mu, sigma)
        obs = nxt["obs"]
    last_values = ac.cri(obs)
    f = ac.flat()
    snap = dict(a=f["actor"].clone(), c=f["critic"].clone())

    def restore():                                     # every timed step starts from the same policy
        f["actor"].copy_(snap["a"])
        f["critic"].copy_(snap["c"])
        for opt in (run.optimizer_actor, run.optimizer_critic):
            opt.m.zero_()
            opt.v.zero_()
            opt.state_dev.zero_()

    def step():
        restore()
        st.step = w["T"]
        run.log_dict = {}
        run.learn(last_values)

    return run, ac, st, last_values, step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="vision", choices=list(WORKLOADS) + ["dagger", "depth2pc"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n-steps", type=int, default=0, help="override the rollout length T (e.g. 128 for the vision workload)")
    ap.add_argument("--student", default="pointnet", choices=["pointnet", "conv3d"],
                    help="dagger workload: PointNet on --points clouds (cfg 5 analogue) or the reference's shipped Conv3DNet config")
    ap.add_argument("--points", type=int, default=4096, help="dagger workload: points per student cloud (BASELINE cfg 5: 4096)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16x6"],
                    help="vision encoder forward: exact fp32 MFMA (default) or the opt-in split-bf16 path")
    args = ap.parse_args()

    from partmanip_amd import dist as pdist, ops
    rank, world, local = pdist.init_from_env("nccl")
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if args.workload == "depth2pc":
        if rank == 0:
            print(json.dumps(run_depth2pc(args, device)))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if args.workload == "dagger":
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):       # the runners print progress lines: keep stdout = ONE JSON line
            out = run_dagger(args, device, rank, world)
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    w = dict(WORKLOADS[args.workload])
    if args.n_steps:
        w["T"] = args.n_steps
        w["name"] = w["name"] + f"_T{args.n_steps}"
    if args.precision != "f32" and w["net"]["name"] == "PointNet":
        w["net"] = dict(w["net"], precision=args.precision)
        w["name"] = w["name"] + "_" + args.precision
    cfg = make_cfg(w, device)
    run, ac, st, last_values, step = build_runner(w, cfg, device, rank)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    dominant = "pointnet_enc_fwd" if args.workload == "vision" else None
    if dominant:
        ops.TIMER.enable(dominant, "pointnet_enc_bwd")
    SA_LEVELS = {"64x64x128": (64, 64, 128, 256), "128x128x256": (128, 128, 256, 64)}     # C1, C2, C3, groups per cloud
    if args.workload == "vision_pn2":
        ops.TIMER.enable(*[f"sa_{d}_{k}" for d in ("fwd", "bwd") for k in SA_LEVELS])
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ops.TIMER.disable()
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = w["N"] * w["T"] * world / (dt / args.steps)

    metric = ("PPO env-steps/sec (whole node), 4096 envs x 1024-pt clouds" if args.workload.startswith("vision")
              else "PPO env-steps/sec (whole node), 4096 envs x 128 steps, state obs (BASELINE cfg 2)")
    out = dict(metric=metric, value=value, unit="env-steps/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f32" if args.precision == "f32" else f"f32 (encoder forward: {args.precision} split MFMA)",
               data="synthetic",
               config=dict(workload=w["name"], envs_per_gpu=w["N"], n_steps=w["T"], points=1024 if args.workload.startswith("vision") else 0,
                           minibatch=2048, n_updates=5, parallelism=f"dp{world}",
                           train_scalars={k: float(v) for k, v in run.log_dict.items() if k.startswith("Train/")}))
    if dominant:
        mean_ms, n_launch = ops.TIMER.mean_ms(dominant)
        flops = 2.0 * ENC_MAC_PER_POINT * 1024 * 2048           # one launch = 2048 clouds x 1024 points
        achieved = flops / (mean_ms * 1e-3) / 1e12
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            traffic = json.load(open(tfile)).get("pn_fwd_kernel_bytes_per_launch")
        out["roofline"] = dict(bound="mfma", kernel="pn_fwd_kernel", achieved=achieved, peak=PEAK_F32_MFMA_TFLOPS,
                               unit="TFLOP/s", frac=achieved / PEAK_F32_MFMA_TFLOPS, traffic=traffic,
                               launches=n_launch, mean_launch_ms=mean_ms, flops_per_launch=flops,
                               note="training forward: also writes the layer-2 activations (1 KB/point) that spare the "
                                    "backward a recompute; 0.83 without that store (DESIGN.md 3.2)")
        bwd = ops.TIMER.mean_ms("pointnet_enc_bwd")
        if bwd:
            out["roofline"]["enc_bwd_mean_ms"] = bwd[0]
            # second kernel of the step: the structured backward EXECUTES 2 dense GEMMs (dW2 = dz2^T h1, dh1 = dz2 W2:
            # 2 x 2*256*128 flops per point); the timer brackets the whole C call (kernel + its five small follow-ups)
            bflops = 2.0 * 2 * 256 * 128 * 1024 * 2048
            out["roofline"]["enc_bwd_executed_tflops"] = bflops / (bwd[0] * 1e-3) / 1e12
            out["roofline"]["enc_bwd_frac"] = out["roofline"]["enc_bwd_executed_tflops"] / PEAK_F32_MFMA_TFLOPS
    if args.workload == "vision_pn2":
        # the four fused set-abstraction kernels; `achieved` counts the MFMA flops each launch EXECUTES
        # (fwd: layers 2-3; bwd: dH2 + dW2 + dH1 -- layer 2 is loaded from what the forward saved) on 2048 clouds
        kern = {}
        for k, (c1, c2, c3, S) in SA_LEVELS.items():
            rows = 2048.0 * S * 32
            for d, macs in (("fwd", c1 * c2 + c2 * c3), ("bwd", 2 * c1 * c2 + c2 * c3)):
                t = ops.TIMER.mean_ms(f"sa_{d}_{k}")
                if t:
                    kern[f"sa_{d}_{k}"] = dict(mean_launch_ms=t[0], launches=t[1], tflops=2 * rows * macs / (t[0] * 1e-3) / 1e12)
        if kern:
            name = max(kern, key=lambda n: kern[n]["mean_launch_ms"] * kern[n]["launches"])
            out["roofline"] = dict(bound="mfma", kernel=name, achieved=kern[name]["tflops"], peak=PEAK_F32_MFMA_TFLOPS,
                                   unit="TFLOP/s", frac=kern[name]["tflops"] / PEAK_F32_MFMA_TFLOPS, traffic=None,
                                   kernels=kern)
    if args.workload == "vision" and args.precision == "f32" and world == 1:
        # the same workload on the opt-in split-bf16 encoder forward (reported next to, not instead of, the fp32 line)
        ac.actor.precision = ac.critic.precision = "bf16x3"
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        fence()
        dt3 = (time.perf_counter() - t1) / 2
        ac.actor.precision = ac.critic.precision = "f32"
        out["optional_paths"] = dict(encoder_forward_bf16x3=dict(
            value=w["N"] * w["T"] / dt3, unit="env-steps/s", ms_per_step=dt3 * 1e3,
            note="pm_pointnet_enc_fwd_bf3: a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on bf16 MFMAs, fp32 accumulate; "
                 "~1e-5 relative; passes the golden vision-PPO cases at the fp32 path's tolerances; backward stays fp32"))
        # ... and on the three-plane split (six bf16 MFMAs per product block): the fp32 kernel's error level
        ac.actor.precision = ac.critic.precision = "bf16x6"
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        fence()
        dt6 = (time.perf_counter() - t1) / 2
        ac.actor.precision = ac.critic.precision = "f32"
        out["optional_paths"]["encoder_forward_bf16x6"] = dict(
            value=w["N"] * w["T"] / dt6, unit="env-steps/s", ms_per_step=dt6 * 1e3,
            note="pm_pointnet_enc_fwd_bf6: operands split into three bf16 planes, products a0b0+a0b1+a1b0+a0b2+a1b1+a2b0 "
                 "on bf16 MFMAs with fp32 accumulate; error against fp64 no larger than the fp32 MFMA kernel's "
                 "(tests/test_gpu_learner.py::test_pointnet_bf16x6_forward_has_fp32_class_error); backward stays fp32")
        # ... with the critic loop issued on a second HIP stream next to the actor loop (bit-identical results; off by
        # default for this workload because concurrent kernels blur the per-kernel timing the roofline block reports)
        run.overlap = True
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        fence()
        dto = (time.perf_counter() - t1) / 2
        run.overlap = False
        out["optional_paths"]["actor_critic_on_two_streams"] = dict(
            value=w["N"] * w["T"] / dto, unit="env-steps/s", ms_per_step=dto * 1e3,
            note="PARTMANIP_OVERLAP=1: same arithmetic, critic step k runs concurrently with actor step k")
        # ... and through the PointNet++ (SSG) plug-in backbone BASELINE.json's config text names: FPS + ball query
        # once per rollout, fused set-abstraction kernels (pm_sa_fwd_f32 / pm_sa_bwd_f32)
        w2 = dict(WORKLOADS["vision_pn2"])
        _, _, _, _, step2 = build_runner(w2, make_cfg(w2, device), device, rank)
        step2()
        fence()
        t1 = time.perf_counter()
        step2()
        fence()
        dt2 = time.perf_counter() - t1
        out["optional_paths"]["pointnet2_ssg_backbone"] = dict(
            value=w2["N"] * w2["T"] / dt2, unit="env-steps/s", ms_per_step=dt2 * 1e3, workload=w2["name"],
            note="same rollouts, network.name = PointNet2 (npoints 256/64, radii 0.2/0.4, 32 samples, mlps 64-64-128 / "
                 "128-128-256 / 256-512); fp32")
        del step2
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = lambda t: t.detach().cpu()
        roll = dict(observations=cpu(st.observations), actions=cpu(st.actions), values=cpu(st.values),
                    returns=cpu(st.returns), actions_log_prob=cpu(st.actions_log_prob), advantages=cpu(st.advantages),
                    mu=cpu(st.mu), sigma=cpu(st.sigma), rewards_full=cpu(st.rewards), values_full=cpu(st.values),
                    dones_full=cpu(st.dones), succs_full=cpu(st.succs), last_values=cpu(last_values))
        sd_cpu = {k: cpu(v).clone() for k, v in ac.state_dict().items()}
        out["cpu_baseline"] = cpu_baseline(w, roll, sd_cpu, cfg)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
