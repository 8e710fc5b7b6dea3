"""Golden-case definitions shared by `make_golden.py` (runs the reference in the
dev container) and the parity tests (which run the oracle / the HIP path).

Pure data + deterministic input builders; nothing here imports the reference.
Hyper-parameter names follow the reference's yaml (cfg/algos/ppo.yaml:1-48,
cfg/algos/dagger_tsdf.yaml:1-38).
"""
import copy
import numpy as np

from .detgen import det_uniform, det_normal, det_bernoulli, det_u01, linear_init

TRICKS_DEFAULT = dict(mini_adv_norm=False, whole_adv_norm=False, use_state_norm=False,
                      use_clipped_value_loss=False, use_grad_clip=True, max_grad_norm=0.5)
TRICKS_ALLON = dict(mini_adv_norm=True, whole_adv_norm=True, use_state_norm=False,
                    use_clipped_value_loss=True, use_grad_clip=True, max_grad_norm=0.5)
TRICKS_NOCLIP = dict(mini_adv_norm=False, whole_adv_norm=False, use_state_norm=False,
                     use_clipped_value_loss=False, use_grad_clip=False, max_grad_norm=0.5)

GAE_CASES = {
    "gae_none":      dict(T=16, N=8, succ_value=None, whole_adv_norm=False, seed=11),
    "gae_500":       dict(T=16, N=8, succ_value=500, whole_adv_norm=False, seed=12),
    "gae_none_norm": dict(T=16, N=8, succ_value=None, whole_adv_norm=True, seed=13),
    "gae_500_norm":  dict(T=16, N=8, succ_value=500, whole_adv_norm=True, seed=14),
    "gae_T1":        dict(T=1, N=5, succ_value=0, whole_adv_norm=False, seed=15),
    "gae_ragged":    dict(T=37, N=131, succ_value=None, whole_adv_norm=True, seed=16),
}

_MLP_NET = dict(name="MLP", hid_dim=[64, 64, 64], activation="tanh")
_PN_NET = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False)
_PN_NET_MAX = dict(name="PointNet", activation="tanh", max_mean=False, sub_mean=False)
_PN_NET_SUB = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=True)


def _ppo(net, N, T, O, n_mb, n_up, tricks, sampler="sequential", succ_value=None,
         lr=3e-3, desired_kl=0.1, lr_schedule="fixed", seed=100, old_noise=0.05, A=10, noise_ramp=False):
    return dict(net=net, N=N, T=T, O=O, A=A, n_minibatches=n_mb, n_updates=n_up, tricks=tricks,
                sampler=sampler, succ_value=succ_value, lr=lr, desired_kl=desired_kl,
                lr_schedule=lr_schedule, seed=seed, old_noise=old_noise, noise_ramp=noise_ramp,
                gamma=0.99, lam=0.95, epsilon_clip=0.2, action_std=0.5, max_iterations=1000, it=7)


PPO_CASES = {
    "ppo_mlp_default": _ppo(_MLP_NET, 8, 16, 32, 4, 2, TRICKS_DEFAULT, seed=101),
    "ppo_mlp_allon":   _ppo(_MLP_NET, 8, 16, 32, 4, 2, TRICKS_ALLON, succ_value=500, seed=102,
                            lr_schedule="linear_decay"),
    "ppo_mlp_noclip":  _ppo(_MLP_NET, 8, 16, 32, 4, 2, TRICKS_NOCLIP, seed=103, lr_schedule="step_decay"),
    "ppo_mlp_random":  _ppo(_MLP_NET, 8, 16, 32, 4, 2, TRICKS_DEFAULT, sampler="random", seed=104),
    # some mini-batches exceed desired_kl and must be skipped (ppo.py:337-338)
    "ppo_mlp_klskip":  _ppo(_MLP_NET, 8, 16, 32, 4, 2, TRICKS_DEFAULT, seed=105, old_noise=0.2, noise_ramp=True, lr=1e-3),
    # drop_last with a ragged tail: 7*9=63 samples, 4 mini-batches of 15, 3 samples dropped
    "ppo_mlp_ragged":  _ppo(_MLP_NET, 7, 9, 19, 4, 3, TRICKS_ALLON, seed=106, A=7),
    "ppo_pn_maxmean":  _ppo(_PN_NET, 4, 4, 3072, 2, 2, TRICKS_DEFAULT, seed=107, lr=2e-4, old_noise=0.02),
    "ppo_pn_max":      _ppo(_PN_NET_MAX, 4, 4, 3072, 2, 2, TRICKS_ALLON, seed=108, lr=2e-4, old_noise=0.02),
    # NOTE sub_mean=True cannot be pinned through ppo.update without proprio: the reference centres IN PLACE on a view of
    # the mini-batch (network.py:172-173), the critic's forward then rewrites the tensor the actor's graph saved, and
    # its backward raises "modified by an inplace operation".  Centring is pinned by DAGGER_CASES["dagger_pn_submean"].
}

# rollout side: Normalization over three (N, O) batches; random_act_cri of an MLP actor-critic (actor_critic.py:36-47)
ROLLOUT_CASE = dict(N=37, O=19, A=6, seed=401, torch_seed=4242, action_std=0.7, max_action=1.5, net=_MLP_NET)

DAGGER_CASES = {
    # student MLP on a 40-d obs, teacher MLP on a 32-d state
    "dagger_mlp": dict(stu_net=_MLP_NET, tea_net=_MLP_NET, N=8, buf_size=6, n_fill=8, O_s=40, O_t=32,
                       proprio=0, A=10, n_minibatches=3, n_updates=2, lr=3e-3, lr_schedule="linear_decay",
                       sampler="random", seed=201, action_std=0.1, max_iterations=1000, it=5, torch_seed=77),
    # student PointNet (+7-d proprio appended), teacher MLP
    # the reference's shipped default student (cfg/algos/dagger_tsdf.yaml): Conv3DNet on a 50^3 volume + proprio
    "dagger_conv3d": dict(stu_net=dict(name="Conv3DNet", activation="tanh"), tea_net=_MLP_NET, N=4, buf_size=5, n_fill=4,
                          O_s=50 ** 3 + 7, O_t=32, proprio=7, A=10, n_minibatches=2, n_updates=2, lr=1e-4,
                          lr_schedule="fixed", sampler="random", seed=203, action_std=0.1, max_iterations=1000, it=5,
                          torch_seed=79),
    "dagger_pn":  dict(stu_net=_PN_NET, tea_net=_MLP_NET, N=4, buf_size=5, n_fill=4, O_s=3072 + 7, O_t=32,
                       proprio=7, A=10, n_minibatches=2, n_updates=2, lr=1e-3, lr_schedule="fixed",
                       sampler="sequential", seed=202, action_std=0.1, max_iterations=1000, it=5, torch_seed=78),
    # the same student with per-cloud centring (network.py:172-173); the clouds get a per-cloud offset below
    "dagger_pn_submean":  dict(stu_net=_PN_NET_SUB, tea_net=_MLP_NET, N=4, buf_size=5, n_fill=4, O_s=3072 + 7, O_t=32,
                       proprio=7, A=10, n_minibatches=2, n_updates=2, lr=1e-3, lr_schedule="fixed",
                       sampler="sequential", seed=204, action_std=0.1, max_iterations=1000, it=5, torch_seed=78),
}


# --------------------------------------------------------------------------- inputs
def gae_inputs(c):
    T, N, s = c["T"], c["N"], c["seed"] * 1000
    rewards = det_normal((T, N, 1), s + 1)
    values = det_normal((T, N, 1), s + 2)
    last_values = det_normal((N, 1), s + 3)
    dones = det_bernoulli((T, N, 1), s + 4, 0.15)
    succs = dones & det_bernoulli((T, N, 1), s + 5, 0.5)
    return dict(rewards=rewards, values=values, last_values=last_values, dones=dones, succs=succs)


def net_param_shapes(net, in_dim, out_dim, proprio=0):
    """[(state_dict key suffix, (out,in))] in module order (network.py:31-41,147-160)."""
    if net["name"] == "MLP":
        dims = [in_dim] + list(net["hid_dim"]) + [out_dim]
        return [(f"model.{2 * i}", (dims[i + 1], dims[i])) for i in range(len(dims) - 1)]
    if net["name"] == "PointNet":
        c = (in_dim - proprio) // 1024
        feat = 512 * (2 if net["max_mean"] else 1) + proprio
        return [("mlp.0", (128, c)), ("mlp.2", (256, 128)), ("mlp.4", (512, 256)),
                ("final_mlp.0", (128, feat)), ("final_mlp.2", (32, 128)), ("final_mlp.4", (out_dim, 32))]
    if net["name"] == "SparseUNet":
        c0, c1, c2 = net.get("channels", [32, 64, 128])
        return [("conv0", (c0, 27 * 4)), ("down0", (c1, 8 * c0)), ("conv1", (c1, 27 * c1)), ("down1", (c2, 8 * c1)),
                ("conv2", (c2, 27 * c2)), ("up1", (c1, c2 + c1)), ("up0", (c0, c1 + c0)),
                ("final_mlp.0", (128, c0 + proprio)), ("final_mlp.2", (32, 128)), ("final_mlp.4", (out_dim, 32))]
    if net["name"] == "PointNet2":
        # partmanip_amd.algo_utils.network.PointNet2: per level Linear(pad4(3 + C_prev), d0)-act-Linear-act-...; the PointNet head
        P = int(net.get("point_num", 1024))
        cf = (in_dim - proprio) // P - 3
        out = []
        for l, dims in enumerate(net.get("mlps", [[64, 64, 128], [128, 128, 256], [256, 512]])):
            cin = (3 + cf + 3) // 4 * 4
            for i, d in enumerate(dims):
                out.append((f"sa.{l}.{2 * i}", (d, cin)))
                cin = d
            cf = dims[-1]
        return out + [("final_mlp.0", (128, cf + proprio)), ("final_mlp.2", (32, 128)), ("final_mlp.4", (out_dim, 32))]
    raise KeyError(net["name"])


def sparse_clouds(B, P, R, seed, n_distinct=None, pad_tail=0):
    """'depth_sparse'-style rows (x, y, z, f): integer voxel coordinates as floats + a feature in (-0.2, 0.2) per VOXEL; the
    last `pad_tail` rows of every cloud repeat voxel (0, 0, 0) the way `TSDFVolume.sparse_voxel` pads short clouds
    (utils/depth2tsdf.py:116-119); n_distinct < P makes further rows repeat earlier voxels."""
    out = np.zeros((B, P, 4), dtype=np.float32)
    n = min(n_distinct or P, P)
    assert n < R * R
    for b in range(B):
        s = seed * 1000 + b * 10
        order = np.argsort(det_u01((R * R,), s), kind="stable")
        order = order[order != 0][:n]                                        # n distinct (x, y) columns; (0, 0) is the padding voxel's
        xs, ys = order // R, order % R
        zs = np.clip((0.4 * xs + 0.3 * ys + 3.0 * det_u01((n,), s + 1)).astype(np.int64), 0, R - 1)   # a noisy tilted surface
        vox = np.stack([xs, ys, zs], 1).astype(np.float32)
        fv = det_uniform((n,), s + 2, -0.2, 0.2)
        idx = np.concatenate([np.arange(n), (det_u01((P - n,), s + 3) * n).astype(np.int64) % n]) if n < P else np.arange(P)
        out[b, :, :3] = vox[idx]
        out[b, :, 3] = fv[idx]
        if pad_tail:
            out[b, P - pad_tail:, :3] = 0.0
            out[b, P - pad_tail:, 3] = 0.125
    return out.reshape(B, P * 4)


def actor_critic_state(net, in_dim, A, action_std, seed, proprio=0):
    """Deterministic initial `ActorCritic.state_dict()` (numpy float32), keys as actor_critic.py:16-22."""
    sd = {"log_std": np.full((A,), np.log(action_std), dtype=np.float32)}
    if net["name"] == "Conv3DNet":
        for which, out_dim, s0 in (("actor", A, seed * 100), ("critic", 1, seed * 100 + 50)):
            for k, v in conv3d_state(dict(proprio=proprio, out=out_dim, seed=s0)).items():
                sd[f"{which}.{k}"] = v
        return sd
    for which, out_dim, s0 in (("actor", A, seed * 100), ("critic", 1, seed * 100 + 50)):
        shapes = net_param_shapes(net, in_dim, out_dim, proprio)
        for li, (key, (o, i)) in enumerate(shapes):
            gain = 1.0
            if li == len(shapes) - 1 and which == "actor":
                gain = 0.3   # small policy head, like the reference's 0.01-gain orthogonal init
            w, b = linear_init(o, i, s0 + li, gain)
            sd[f"{which}.{key}.weight"] = w
            sd[f"{which}.{key}.bias"] = b
    return sd


def ppo_raw_inputs(c):
    """Rollout tensors that do not depend on the policy (obs, actions, rewards, masks)."""
    T, N, O, A, s = c["T"], c["N"], c["O"], c["A"], c["seed"] * 1000
    if c["net"]["name"] == "PointNet":
        pts = det_uniform((T, N, 1024, O // 1024), s + 1, -1.0, 1.0)
        shift = det_uniform((T, N, 1, O // 1024), s + 9, -0.5, 0.5)
        obs = (pts + shift).reshape(T, N, O).astype(np.float32)
    else:
        obs = det_normal((T, N, O), s + 1)
    actions = det_uniform((T, N, A), s + 2, -0.999, 0.999)
    actions.reshape(-1)[::17] = 1.0      # saturated actions exercise the atanh clamp (actor_critic.py:95)
    actions.reshape(-1)[5::23] = -1.0
    rewards = det_normal((T, N, 1), s + 3)
    dones = det_bernoulli((T, N, 1), s + 4, 0.1)
    succs = dones & det_bernoulli((T, N, 1), s + 5, 0.5)
    noise = dict(values=det_normal((T, N, 1), s + 6), logp=det_normal((T, N, 1), s + 7),
                 mu=det_normal((T, N, A), s + 8), sigma=det_normal((T, N, A), s + 10),
                 last_values=det_normal((N, 1), s + 11))
    return dict(observations=obs, actions=actions, rewards=rewards, dones=dones, succs=succs, noise=noise)


def dagger_raw_inputs(c):
    s = c["seed"] * 1000
    N, nf = c["N"], c["n_fill"]
    stu, tea = [], []
    for k in range(nf):
        if c["stu_net"]["name"] == "PointNet":
            npc = c["O_s"] - c["proprio"]
            pts = det_uniform((N, 1024, npc // 1024), s + 10 * k + 1, -1.0, 1.0)
            if c["stu_net"].get("sub_mean"):                     # a per-cloud offset, so that centring matters
                pts = pts + det_uniform((N, 1, npc // 1024), s + 10 * k + 7, -0.5, 0.5)
            pts = pts.reshape(N, npc)
            pro = det_normal((N, c["proprio"]), s + 10 * k + 2)
            stu.append(np.concatenate([pts, pro], axis=1).astype(np.float32))
        elif c["stu_net"]["name"] == "Conv3DNet":
            stu.append(np.concatenate([det_uniform((N, c["O_s"] - c["proprio"]), s + 10 * k + 1, -1.0, 1.0),
                                       det_normal((N, c["proprio"]), s + 10 * k + 2)], axis=1).astype(np.float32))
        else:
            stu.append(det_normal((N, c["O_s"]), s + 10 * k + 1))
        tea.append(det_normal((N, c["O_t"]), s + 10 * k + 3))
    return dict(stu=stu, tea=tea)


def case_copy(c):
    return copy.deepcopy(c)


# ---- observation side: depth images -> world cloud (utils/depth2tsdf.py:136-173) ---------------------------------
DEPTH2PC_CASES = {
    # 2 envs x 3 views of 24 x 32 pixels; a workspace box that cuts roughly half of the points away
    "depth2pc_small": dict(b=2, m=3, h=24, w=32, K=64, size=0.5, vol_origin=[-0.25, -0.25, 0.35], seed=301,
                           intr=[[30.0, 0.0, 15.5], [0.0, 28.0, 11.5], [0.0, 0.0, 1.0]]),
}


def depth2pc_inputs(c):
    from .detgen import det_uniform
    depth = det_uniform((c["b"], c["m"], c["h"], c["w"]), c["seed"], 0.3, 1.0)
    depth[:, :, :2, :] = 0.0                                     # invalid depth rows -> points at the camera centre
    pose = np.zeros((c["m"], 4, 4), dtype=np.float32)
    ang = det_uniform((c["m"],), c["seed"] + 1, -0.6, 0.6)
    for i in range(c["m"]):
        ca, sa = np.cos(ang[i]), np.sin(ang[i])
        cb, sb = np.cos(0.5 * ang[i] + 0.2), np.sin(0.5 * ang[i] + 0.2)
        rz = np.array([[ca, -sa, 0.0], [sa, ca, 0.0], [0.0, 0.0, 1.0]])
        rx = np.array([[1.0, 0.0, 0.0], [0.0, cb, -sb], [0.0, sb, cb]])
        pose[i, :3, :3] = (rz @ rx).astype(np.float32)              # a general rotation: no zero entries
        pose[i, :3, 3] = det_uniform((3,), c["seed"] + 2 + i, -0.1, 0.1)
        pose[i, 3, 3] = 1.0
    return dict(depth=depth, cam_pose=pose)


# ---- behaviour cloning from offline shards (algorithms/bc.py) -----------------------------------------------------
BC_CASES = {
    # one scene x 22 steps (a single scene keeps the index -> file map independent of os.listdir order);
    # batch = 22 // 4 = 5 -> 5 mini-batches per epoch, the last one ragged with 2 rows (DataLoader keeps it; a
    # 1-row tail would crash the reference itself: bc.py:128 squeezes the batch axis away)
    "bc_mlp": dict(net=_MLP_NET, n_steps=22, D=40, S=5, A=10, n_minibatches=4, max_iterations=3, lr=3e-3,
                   lr_schedule="linear_decay", seed=401, action_std=0.1, torch_seed=91),
}


def bc_dataset(c):
    """rows of the offline shards: tsdf (n, D) ~N(0,1), action (n, A) in (-1, 1), proprio_state (n, S)."""
    n = c["n_steps"]
    return dict(tsdf=det_normal((n, c["D"]), c["seed"]), action=det_uniform((n, c["A"]), c["seed"] + 1, -0.9, 0.9),
                proprio_state=det_normal((n, c["S"]), c["seed"] + 2))


def bc_write_dataset(c, folder):
    import os
    d = bc_dataset(c)
    scene = os.path.join(folder, "scene_00000")
    os.makedirs(scene, exist_ok=True)
    for i in range(c["n_steps"]):
        np.save(os.path.join(scene, f"step_{str(i).zfill(5)}.npy"),
                dict(tsdf=d["tsdf"][i], action=d["action"][i], proprio_state=d["proprio_state"][i]), allow_pickle=True)
    return d


# ---- mixed BC + on-policy DAgger: offline shards preloaded into the ring (storage.py:58-82, dagger.py:186-187) ------
# 2 scenes x 8 steps = 16 offline rows, then 3 on-policy env steps of 4 rows: 28 rows through a 20-row ring (wraps;
# the reference itself crashes when the preload leaves the write index off a multiple of num_envs at the wrap).
# The student is an MLP on [tsdf(40) | proprio(6)] rows (the loader only flattens `tsdf` and appends `proprio_state`).
DAGGER_OFFLINE_CASE = dict(stu_net=_MLP_NET, tea_net=_MLP_NET, N=4, buf_size=5, n_fill=3, D=40, proprio=6, O_t=32, A=10,
                           scenes=2, steps=8, n_minibatches=2, n_updates=2, lr=2e-3, lr_schedule="linear_decay",
                           sampler="random", seed=701, action_std=0.1, max_iterations=1000, it=7, torch_seed=83)


def dagger_offline_rows(c):
    n = c["scenes"] * c["steps"]
    return dict(tsdf=det_normal((n, c["D"]), c["seed"]), proprio_state=det_normal((n, c["proprio"]), c["seed"] + 1),
                tea_obs=det_normal((n, c["O_t"]), c["seed"] + 2))


def dagger_offline_write(c, folder):
    """`folder/scene_XXXXX/step_XXXXX.npy` dict shards ({tsdf, proprio_state, tea_obs}) in (scene, step) order."""
    import os
    d = dagger_offline_rows(c)
    for sc in range(c["scenes"]):
        scene = os.path.join(folder, f"scene_{str(sc).zfill(5)}")
        os.makedirs(scene, exist_ok=True)
        for st in range(c["steps"]):
            i = sc * c["steps"] + st
            np.save(os.path.join(scene, f"step_{str(st).zfill(5)}.npy"),
                    dict(tsdf=d["tsdf"][i].reshape(5, 8), proprio_state=d["proprio_state"][i], tea_obs=d["tea_obs"][i]),
                    allow_pickle=True)
    return d


def dagger_offline_online(c):
    """The on-policy rows added after the preload: n_fill env steps of (N, D + proprio) / (N, O_t)."""
    s = c["seed"] * 1000
    return dict(stu=[det_normal((c["N"], c["D"] + c["proprio"]), s + 10 * k + 1) for k in range(c["n_fill"])],
                tea=[det_normal((c["N"], c["O_t"]), s + 10 * k + 3) for k in range(c["n_fill"])])


# ---- Conv3D TSDF student (network.py:67-94) -----------------------------------------------------------------------
CONV3D_CASES = {
    "conv3d_proprio": dict(B=3, res=50, proprio=5, out=10, seed=501),
    "conv3d_plain": dict(B=2, res=50, proprio=0, out=1, seed=502),
}


def conv3d_state(c):
    """Deterministic `Conv3DNet.state_dict()` (keys of network.py:71-79,119-121), fan-in scaled uniforms."""
    sd = {}
    chans, kern = [1, 16, 32, 32], [5, 3, 3]
    for i in range(3):
        fan = chans[i] * kern[i] ** 3
        w, b = linear_init(chans[i + 1], fan, c["seed"] * 10 + i, 1.7)
        sd[f"encoder.conv{i + 1}.weight"] = w.reshape(chans[i + 1], chans[i], kern[i], kern[i], kern[i])
        sd[f"encoder.conv{i + 1}.bias"] = b
    for j, (o, i_) in zip((0, 2), ((256, 32 * 27 + c["proprio"]), (c["out"], 256))):
        w, b = linear_init(o, i_, c["seed"] * 10 + 5 + j, 1.7)
        sd[f"final_mlp.{j}.weight"], sd[f"final_mlp.{j}.bias"] = w, b
    return sd


def conv3d_inputs(c):
    """TSDF-like volumes in [-1, 1] (+ proprio tail) and the upstream gradient."""
    x = det_uniform((c["B"], c["res"] ** 3 + c["proprio"]), c["seed"] + 100, -1.0, 1.0)
    dy = det_normal((c["B"], c["out"]), c["seed"] + 200)
    return dict(x=x, dy=dy)
