"""CPU oracle for the PPO / DAgger learner hot path.  TEST INFRASTRUCTURE ONLY.

This is a from-scratch restatement (plain PyTorch on CPU, fp32) of the
algorithm the reference implements on the path BASELINE.json names.  It is the
checker for `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg, and nothing else: the product package `partmanip_amd`
never imports it (tests/test_no_oracle_in_product.py enforces that).

Parity status: PINNED for GAE, PPO update (MLP and PointNet backbones), the
actor-critic heads, the mini-batch sampler, the DAgger update (MLP, PointNet and
Conv3DNet students), the `bc` runner, the Conv3DNet module (outputs + parameter
gradients), `TSDFVolume.depth2pc`'s world cloud, `TSDFVolume.integrate` and
`TSDFVolume.sparse_voxel` (everything around its pytorch3d call), and the rollout
side (`Normalization`, `ActorCritic.random_act_cri`) --
each is checked in tests/test_oracle_golden.py against fixtures produced by
running the reference itself (tests/golden/make_golden.py).  The point-set operators
at the bottom (farthest point sampling, ball query, grouping, PointNet++ set
abstraction) have NO implementation in the reference tree (SURVEY.md §8a A15,
A16: the only FPS is a call into un-vendored, unpinned pytorch3d) -- for those
this file is "parity unpinned": it restates the published algorithms
(pytorch3d `sample_farthest_points` defaults; Qi et al. PointNet++ ball query).

Each function cites the reference lines it follows (paths relative to
/root/reference).

The restatement is plain torch and device-agnostic: the whole-update parity tests
(tests/test_gpu_wholeupdate.py) evaluate it on the MI355X through ATen in fp32 and fp64,
where the host cores would take hours.  Three evaluation aids exist for that and change no
mathematics: `grad_chunk` (a mini-batch's mean loss and gradient as size-weighted sums over
row chunks), `_LinearSlabs` (a Linear whose weight gradient is summed over row slabs) and the
batched index builders (`fps_torch`, `ball_query_torch`, `sparse_unet_geometry_torch`); each
is pinned to its plain form on CPU (tests/test_oracle_golden.py, test_oracle_pointnet2_rows.py,
test_oracle_sparse_unet.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LN2PI = math.log(2.0 * math.pi)


# =============================================================================
# storage.py
# =============================================================================
def gae_returns(rewards, values, dones, succs, last_values, gamma, lam, succ_value=None,
                whole_adv_norm=False):
    """algorithms/algo_utils/storage.py:96-114 (`RolloutStorage.compute_returns`).

    rewards/values (T,N,1) f32, dones/succs (T,N,1) bool, last_values (N,1).
    Returns (returns, advantages), both (T,N,1) f32.
    """
    T = rewards.shape[0]
    returns = torch.zeros_like(rewards)
    advantage = 0
    for step in reversed(range(T)):
        next_values = last_values if step == T - 1 else values[step + 1]
        not_terminal = ~dones[step]
        delta = rewards[step] + gamma * next_values - values[step]
        advantage = not_terminal * (delta + gamma * lam * advantage)
        if succ_value is not None:
            returns[step] = (~succs[step]) * (advantage + values[step]) + succs[step] * succ_value
        else:
            returns[step] = advantage + values[step]
    advantages = returns - values
    if whole_adv_norm:
        advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    return returns, advantages


def minibatch_size(cur_buf_size, num_mini_batches):
    """storage.py:126-127: the 2048 cap."""
    return min(int(cur_buf_size // num_mini_batches), 2048)


def minibatch_index_lists(cur_buf_size, num_mini_batches, sampler):
    """One pass over storage.py:125-138's BatchSampler (drop_last=True).

    'random' draws `torch.randperm(n)` from the GLOBAL torch RNG exactly like
    SubsetRandomSampler.__iter__ does.
    """
    mb = minibatch_size(cur_buf_size, num_mini_batches)
    if sampler == "sequential":
        order = list(range(cur_buf_size))
    elif sampler == "random":
        order = torch.randperm(cur_buf_size).tolist()
    else:
        raise ValueError(sampler)
    return [order[i * mb:(i + 1) * mb] for i in range(cur_buf_size // mb)] if mb > 0 else []


def dagger_ring_insert(ring_obs, ring_tea, mix_buf_ind, cur_buf_size, stu_obs, tea_obs):
    """storage.py:84-91 (`add_transitions_dagger`); returns new (mix_buf_ind, cur_buf_size)."""
    n = stu_obs.shape[0]
    ring_obs[mix_buf_ind:mix_buf_ind + n].copy_(stu_obs)
    ring_tea[mix_buf_ind:mix_buf_ind + n].copy_(tea_obs)
    cap = ring_obs.shape[0]
    mix_buf_ind = (mix_buf_ind + n) % cap
    if cur_buf_size < cap:
        cur_buf_size += n
    return mix_buf_ind, cur_buf_size


# =============================================================================
# network.py
# =============================================================================
def _act(name, x):
    if name == "tanh":
        return torch.tanh(x)
    if name in ("relu", "crelu"):
        return torch.relu(x)
    if name == "elu":
        return torch.nn.functional.elu(x)
    if name == "selu":
        return torch.selu(x)
    if name == "lrelu":
        return torch.nn.functional.leaky_relu(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise ValueError(name)


class _LinearSlabs(torch.autograd.Function):
    """`F.linear` whose WEIGHT GRADIENT is evaluated in slabs of the row reduction -- dW = sum_s dY_s^T X_s as one batched product
    -- the same sums in another association.  Used only for >= 2^16 rows on a device (`_lin`): the vendor library runs the
    (64 x 2 M) x (2 M x 64) fp64 product of a PointNet++ level's weight gradient on ONE work-group (105 ms per call, 98 % of the
    restatement's time on the MI355X: tests/test_gpu_wholeupdate.py).  tests/test_oracle_golden.py pins it to autograd's F.linear."""
    SLABS = 256

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        R, S = x2.shape[0], _LinearSlabs.SLABS
        n = R // S * S
        dw = torch.bmm(dy2[:n].reshape(S, n // S, -1).transpose(1, 2), x2[:n].reshape(S, n // S, -1)).sum(0)
        if n < R:
            dw = dw + dy2[n:].t() @ x2[n:]
        return (dy2 @ w).reshape(x.shape), dw, dy2.sum(0)


def _lin(p, prefix, x):
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    if x.device.type != "cpu" and x.numel() // x.shape[-1] >= 65536:
        return _LinearSlabs.apply(x, w, b)
    return torch.nn.functional.linear(x, w, b)


def mlp_forward(p, prefix, net_cfg, x):
    """network.py:27-54: Linear-act-...-Linear (no act after the last)."""
    n = len(net_cfg["hid_dim"]) + 1
    for i in range(n):
        x = _lin(p, f"{prefix}.model.{2 * i}", x)
        if i < n - 1:
            x = _act(net_cfg["activation"], x)
    return x


def pointnet_forward(p, prefix, net_cfg, x, proprio_shape=0, point_num=1024, argmax_override=None):
    """network.py:165-198.  `sub_mean` re-centres xyz per cloud (out of place here;
    the reference does it in place on its input view, network.py:172-173).

    `argmax_override` (B,512) int64 is a TEST HOOK: the max-pool then gathers those point
    indices instead of taking torch.max's.  Max-pooling is only piecewise differentiable; two
    fp32 implementations with different summation orders legitimately pick different points for
    near-tied channels, so gradient-parity tests pin the pooling index and compare the rest."""
    B = x.shape[0]
    if proprio_shape != 0:
        proprio = x[:, -proprio_shape:]
        pc = x[:, :-proprio_shape].reshape(B, point_num, -1)
    else:
        pc = x.reshape(B, point_num, -1)
    if net_cfg.get("sub_mean", False):
        pc = torch.cat([pc[..., :3] - pc[..., :3].mean(dim=1, keepdim=True), pc[..., 3:]], dim=-1)
    a = net_cfg["activation"]
    h = _act(a, _lin(p, f"{prefix}.mlp.0", pc))
    h = _act(a, _lin(p, f"{prefix}.mlp.2", h))
    h = _lin(p, f"{prefix}.mlp.4", h)
    hmax = h.max(dim=1)[0] if argmax_override is None else torch.gather(h, 1, argmax_override.unsqueeze(1)).squeeze(1)
    if net_cfg["max_mean"]:
        f = torch.cat((hmax, h.mean(dim=1)), dim=-1)
    else:
        f = hmax
    if proprio_shape != 0:
        f = torch.cat((f, proprio), dim=-1)
    f = _act(a, _lin(p, f"{prefix}.final_mlp.0", f))
    f = _act(a, _lin(p, f"{prefix}.final_mlp.2", f))
    return _lin(p, f"{prefix}.final_mlp.4", f)


def conv3dnet_forward(p, prefix, net_cfg, x, proprio_shape=0):
    """network.py:67-94 `Conv3DNet.forward` with `Encoder` :116-139: three strided Conv3d (k 5/3/3, stride 3/3/2,
    padding k//2), the activation after each, channels-first flatten, [+ proprio], Linear-act-Linear."""
    B = x.shape[0]
    n = x.shape[1] - proprio_shape
    res = round(n ** (1 / 3))
    act = net_cfg["activation"]
    h = x[:, :n].reshape(B, 1, res, res, res)
    for i, (k, st) in enumerate(((5, 3), (3, 3), (3, 2))):
        h = _act(act, F.conv3d(h, p[f"{prefix}.encoder.conv{i + 1}.weight"], p[f"{prefix}.encoder.conv{i + 1}.bias"],
                               stride=st, padding=k // 2))
    h = h.reshape(B, -1)
    if proprio_shape != 0:
        h = torch.cat((h, x[:, -proprio_shape:]), dim=-1)
    h = _act(act, _lin(p, f"{prefix}.final_mlp.0", h))
    return _lin(p, f"{prefix}.final_mlp.2", h)


def net_forward(p, prefix, net_cfg, x, proprio_shape=0, geom=None):
    """actor_critic.py:16,19: backbone chosen by `net_cfg['name']`.  geom: PointNet2's neighbourhood tables of these rows
    when the caller built them ahead (`pointnet2_geometry`)."""
    if net_cfg["name"] == "MLP":
        return mlp_forward(p, prefix, net_cfg, x)
    if net_cfg["name"] == "PointNet":
        return pointnet_forward(p, prefix, net_cfg, x, proprio_shape, point_num=int(net_cfg.get("point_num", 1024)))
    if net_cfg["name"] == "PointNet2":
        return pointnet2_forward(p, prefix, net_cfg, x, proprio_shape, geom=geom)
    if net_cfg["name"] == "Conv3DNet":
        return conv3dnet_forward(p, prefix, net_cfg, x, proprio_shape)
    if net_cfg["name"] == "SparseUNet":
        return sparse_unet_forward(p, prefix, net_cfg, x, proprio_shape)
    raise ValueError(net_cfg["name"])


# =============================================================================
# actor_critic.py
# =============================================================================
def action_activation(a, activate, max_action):
    """actor_critic.py:84-91."""
    return torch.tanh(a) * max_action if activate == "tanh" else a


def action_deactivation(a, activate, max_action):
    """actor_critic.py:93-100."""
    if activate == "tanh":
        return torch.atanh(torch.clamp(a / max_action, max=1 - 1e-5, min=-1 + 1e-5))
    return a


def gaussian_logp_entropy(mu, log_std, x):
    """actor_critic.py:74-78: MultivariateNormal(mu, scale_tril=diag(exp(log_std)^2)).

    The Cholesky factor handed over is sigma^2, so the effective per-dim std is
    sigma^2 = exp(2*log_std) (verified against the reference
    in tests/test_oracle_golden.py via the fwd_logp/fwd_entropy fixtures).
    """
    s = torch.exp(log_std) * torch.exp(log_std)
    z = (x - mu) / s
    A = mu.shape[-1]
    logp = -0.5 * (z * z).sum(-1) - torch.log(s).sum() - 0.5 * A * LN2PI
    ent = 0.5 * A * (1.0 + LN2PI) + torch.log(s).sum()
    return logp, ent.expand(mu.shape[0])


def update_act_cri(p, model_cfg, obs, actions, proprio_shape=0, with_value=True, geom=None):
    """actor_critic.py:71-82 -> (log_prob (B,), entropy (B,), value (B,1), mu (B,A), log_std rows (B,A)).
    with_value=False skips the critic forward the reference also runs (its result never enters the actor loss): only for the
    tests' fp64 evaluation of the actor's trajectory, where it halves the cost."""
    net = model_cfg["network"]
    mu = net_forward(p, "actor", net, obs, proprio_shape, geom)
    x = action_deactivation(actions, model_cfg["action_activate"], model_cfg["clipAction"])
    logp, ent = gaussian_logp_entropy(mu, p["log_std"], x)
    value = net_forward(p, "critic", net, obs, proprio_shape, geom) if with_value else None
    return logp, ent, value, mu, p["log_std"].repeat(mu.shape[0], 1)


def act(p, model_cfg, obs, proprio_shape=0):
    """actor_critic.py:58-60 (deterministic, tanh-squashed)."""
    mu = net_forward(p, "actor", model_cfg["network"], obs, proprio_shape)
    return action_activation(mu.detach(), model_cfg["action_activate"], model_cfg["clipAction"])


# =============================================================================
# optimiser pieces the reference takes from torch (restated explicitly)
# =============================================================================
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ as called at ppo.py:351,381 (L2, eps 1e-6)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


class Adam:
    """torch.optim.Adam defaults (betas .9/.999, eps 1e-8, no weight decay), single-tensor
    CPU path: lerp for m, addcmul for v, `denom = sqrt(v)/sqrt(bc2) + eps`, step lr/bc1."""

    def __init__(self, params, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps = list(params), lr, b1, b2, eps
        self.m = [torch.zeros_like(q) for q in self.params]
        self.v = [torch.zeros_like(q) for q in self.params]
        self.t = [0 for _ in self.params]
        self.lrs = [lr for _ in self.params]      # per-param lr (param groups)

    def step(self, grads):
        with torch.no_grad():
            for i, (q, g) in enumerate(zip(self.params, grads)):
                if g is None:
                    continue
                self.t[i] += 1
                t = self.t[i]
                self.m[i].lerp_(g, 1 - self.b1)
                self.v[i].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                bc1 = 1 - self.b1 ** t
                bc2 = 1 - self.b2 ** t
                denom = (self.v[i].sqrt() / math.sqrt(bc2)).add_(self.eps)
                q.addcdiv_(self.m[i], denom, value=-(self.lrs[i] / bc1))


# =============================================================================
# ppo.py
# =============================================================================
def actor_loss_terms(logp, mu, log_std_rows, old_logp, adv, old_mu, old_sigma, eps_clip, mini_adv_norm):
    """ppo.py:327-344: (kl_mean, surrogate_loss) on one mini-batch."""
    if mini_adv_norm:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    kl = torch.sum(log_std_rows - old_sigma + (torch.square(old_sigma.exp()) + torch.square(old_mu - mu))
                   / (2.0 * torch.square(log_std_rows.exp())) - 0.5, axis=-1)
    kl_mean = torch.mean(kl)
    ratio = torch.exp(logp - torch.squeeze(old_logp))
    s1 = -torch.squeeze(adv) * ratio
    s2 = -torch.squeeze(adv) * torch.clamp(ratio, 1.0 - eps_clip, 1.0 + eps_clip)
    return kl_mean, torch.max(s1, s2).mean()


def value_loss_fn(value, returns, old_values, eps_clip, clipped):
    """ppo.py:368-374."""
    if clipped:
        with torch.no_grad():
            d = (eps_clip * old_values).abs().mean()
            tgt = old_values + (returns - old_values).clamp(-d, d)
        return (value - tgt).pow(2).mean()
    return (returns - value).pow(2).mean()


def split_params(p):
    actor = [k for k in p if k.startswith("actor.")]
    critic = [k for k in p if k.startswith("critic.")]
    return actor, critic


def _chunks(n, chunk):
    return [(lo, min(lo + chunk, n)) for lo in range(0, n, chunk)]


def ppo_update(p, st, cfg, it, opt=None, grad_sync=None, loops=("actor", "critic"), geom=None, grad_chunk=None):
    """ppo.py:307-411 on explicit tensors.

    p   : dict name -> leaf tensor (ActorCritic.state_dict() layout), updated in place.
    st  : dict with flat-able rollout tensors (observations, actions, values, returns,
          actions_log_prob, advantages, mu, sigma), shapes (T,N,.).
    cfg : ppo.yaml-style dict (n_updates, n_minibatches, sampler, desired_kl, epsilon_clip,
          tricks, lr, lr_schedule, max_iterations, model).
    opt : optional (adam_actor, adam_critic) to continue from; created if None.
    grad_sync : optional callable(list_of_grads, list_of_scalars)->None used by the
          multi-process parity tests to average grads/scalars across ranks.
    loops : which of the two loops to run (tests that need an fp64 evaluation of the actor's trajectory alone).
    grad_chunk : evaluate every mini-batch in row chunks of this size -- the mini-batch's mean loss (and KL) is the size-weighted
          sum of the chunk means and its gradient the same sum of chunk gradients; batch statistics (mini_adv_norm, the clipped
          value loss's width) are taken over the whole mini-batch first.  Same mathematics as one shot (None); for evaluation
          where a whole 2048-cloud mini-batch's intermediates exceed what the tensor library handles (the whole-update GPU tests).
    geom : optional PointNet2 tables of ALL rollout rows (`pointnet2_geometry` of the flat observations): every mini-batch
          takes its rows of them instead of sampling / querying again (they depend on the coordinates only).
    Returns dict(log=..., loss_trace=[...], opt=(adam_actor, adam_critic)).
    """
    tricks, model_cfg = cfg["tricks"], cfg["model"]
    ak, ck = split_params(p)
    for k in p:
        p[k].requires_grad_(True)
    if opt is None:
        opt = (Adam([p[k] for k in ak] + [p["log_std"]], cfg["lr"]), Adam([p[k] for k in ck], cfg["lr"]))
    adam_a, adam_c = opt
    flat = {k: v.reshape(-1, v.shape[-1]) for k, v in st.items()}
    n = flat["observations"].shape[0]
    sampler = cfg["sampler"]
    seq_lists = minibatch_index_lists(n, cfg["n_minibatches"], "sequential") if sampler == "sequential" else None
    trace, sum_surr, sum_kl, kl_max, count, sum_v, n_v = [], 0.0, 0.0, 0.0, 0, 0.0, 0

    def rows_geom(idx):
        if geom is None:
            return None
        r = torch.as_tensor(idx, device=geom[0][0].device)
        return [(c[r], g[r]) for c, g in geom]

    for _ in range(cfg["n_updates"] if "actor" in loops else 0):
        lists = seq_lists if seq_lists is not None else minibatch_index_lists(n, cfg["n_minibatches"], sampler)
        for idx in lists:
            gp = [p[k] for k in ak] + [p["log_std"]]
            if grad_chunk is None:
                logp, _, _, mu, ls = update_act_cri(p, model_cfg, flat["observations"][idx], flat["actions"][idx],
                                                    cfg.get("proprio_shape", 0), with_value="critic" in loops, geom=rows_geom(idx))
                kl_mean, loss = actor_loss_terms(logp, mu, ls, flat["actions_log_prob"][idx], flat["advantages"][idx],
                                                 flat["mu"][idx], flat["sigma"][idx], cfg["epsilon_clip"],
                                                 tricks["mini_adv_norm"])
                grads = None
            else:
                adv = flat["advantages"][idx]
                if tricks["mini_adv_norm"]:
                    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
                kl_mean, loss, grads = 0.0, 0.0, None
                for lo, hi in _chunks(len(idx), grad_chunk):
                    ci, w = idx[lo:hi], (hi - lo) / len(idx)
                    logp, _, _, mu, ls = update_act_cri(p, model_cfg, flat["observations"][ci], flat["actions"][ci],
                                                        cfg.get("proprio_shape", 0), with_value=False, geom=rows_geom(ci))
                    kl_c, loss_c = actor_loss_terms(logp, mu, ls, flat["actions_log_prob"][ci], adv[lo:hi], flat["mu"][ci],
                                                    flat["sigma"][ci], cfg["epsilon_clip"], False)
                    g_c = torch.autograd.grad(loss_c * w, gp)
                    grads = [g.clone() for g in g_c] if grads is None else [a + g for a, g in zip(grads, g_c)]
                    kl_mean, loss = kl_mean + kl_c.detach() * w, loss + loss_c.detach() * w
            kl_mean = kl_mean.detach()
            if grad_sync is not None:
                kl_mean = grad_sync.mean_scalar(kl_mean)
            if float(kl_mean) > kl_max:
                kl_max = float(kl_mean)
            if float(kl_mean) > cfg["desired_kl"]:
                continue
            if grads is None:
                grads = list(torch.autograd.grad(loss, gp))
            if grad_sync is not None:
                grad_sync.mean_grads(grads)
            if tricks["use_grad_clip"]:
                clip_grad_norm(grads[:-1], tricks["max_grad_norm"])     # log_std is NOT clipped (ppo.py:351)
            adam_a.step(grads)
            trace.append(float(loss.detach()))
            sum_surr += float(loss.detach())
            sum_kl += float(kl_mean)
            count += 1

    for _ in range(cfg["n_updates"] if "critic" in loops else 0):
        lists = seq_lists if seq_lists is not None else minibatch_index_lists(n, cfg["n_minibatches"], sampler)
        for idx in lists:
            # ppo.py:366: the critic loop also calls update_act_cri, i.e. runs BOTH networks forward
            if grad_chunk is None:
                _, _, value, _, _ = update_act_cri(p, model_cfg, flat["observations"][idx], flat["actions"][idx],
                                                   cfg.get("proprio_shape", 0), geom=rows_geom(idx))
                loss = value_loss_fn(value, flat["returns"][idx], flat["values"][idx], cfg["epsilon_clip"],
                                     tricks["use_clipped_value_loss"])
                grads = list(torch.autograd.grad(loss, [p[k] for k in ck]))
            else:
                ret, old_v = flat["returns"][idx], flat["values"][idx]
                if tricks["use_clipped_value_loss"]:                 # the clip width is a mean over the WHOLE mini-batch (ppo.py:370)
                    d = (cfg["epsilon_clip"] * old_v).abs().mean()
                    ret = old_v + (ret - old_v).clamp(-d, d)
                loss, grads = 0.0, None
                for lo, hi in _chunks(len(idx), grad_chunk):
                    ci, w = idx[lo:hi], (hi - lo) / len(idx)
                    value = net_forward(p, "critic", model_cfg["network"], flat["observations"][ci], cfg.get("proprio_shape", 0), rows_geom(ci))
                    loss_c = (ret[lo:hi] - value).pow(2).mean()
                    g_c = torch.autograd.grad(loss_c * w, [p[k] for k in ck])
                    grads = [g.clone() for g in g_c] if grads is None else [a + g for a, g in zip(grads, g_c)]
                    loss = loss + loss_c.detach() * w
            if grad_sync is not None:
                grad_sync.mean_grads(grads)
            if tricks["use_grad_clip"]:
                clip_grad_norm(grads, tricks["max_grad_norm"])
            adam_c.step(grads)
            trace.append(float(loss.detach()))
            sum_v += float(loss.detach())
            n_v += 1

    lr_now = adam_a.lrs[0]
    if cfg["lr_schedule"] == "linear_decay":
        lr_now = max(cfg["lr"] * (1 - it / cfg["max_iterations"]), 1e-5)
    elif cfg["lr_schedule"] == "step_decay":
        lr_now = 1e-5 if it > cfg["max_iterations"] // 2 else cfg["lr"]
    adam_a.lrs = [lr_now for _ in adam_a.lrs]                     # actor optimiser only (ppo.py:392,399)
    log = {
        "Train/value_gt_return_mean": float(st["returns"].mean()),
        "Train/value_gt_return_max": float(st["returns"].max()),
        "Train/learning_rate": lr_now,
        "Train/value_function_loss": sum_v / n_v if "critic" in loops else float("nan"),
        "Train/surrogate_loss": sum_surr / count if "actor" in loops else float("nan"),   # ZeroDivisionError if every mb was skipped, as ppo.py:387
        "Train/kl": sum_kl / count if "actor" in loops else float("nan"),
        "Train/kl_max": kl_max,
        "Train/kl_update_count": count,
    }
    for k in p:
        p[k].requires_grad_(False)
    return dict(log=log, loss_trace=trace, opt=opt)


# =============================================================================
# dagger.py
# =============================================================================
def dagger_update(stu, tea, ring_obs, ring_tea, cur_buf_size, cfg, it, opt=None, grad_sync=None, grad_chunk=None):
    """dagger.py:299-337.  `stu`/`tea`: state dicts; cfg has model (student), tea_model,
    n_updates, n_minibatches, sampler, lr, lr_schedule, max_iterations, proprio_shape.
    grad_sync: as in ppo_update (multi-process parity tests average the gradients across ranks); grad_chunk: as in ppo_update
    (the mini-batch's mean loss and its gradient as size-weighted sums over row chunks)."""
    if cur_buf_size < 16:
        return None
    for k in stu:
        stu[k].requires_grad_(True)
    names = list(stu.keys())
    # dagger.py:56: Adam over student.parameters() = log_std, actor.*, critic.* (module order)
    if opt is None:
        opt = Adam([stu[k] for k in names], cfg["lr"])
    trace = []
    for _ in range(cfg["n_updates"]):
        for idx in minibatch_index_lists(cur_buf_size, cfg["n_minibatches"], cfg["sampler"]):
            with torch.no_grad():
                tea_act = act(tea, cfg["tea_model"], ring_tea[idx])
            if grad_chunk is None:
                mu = net_forward(stu, "actor", cfg["model"]["network"], ring_obs[idx], cfg.get("proprio_shape", 0))
                stu_act = action_activation(mu, cfg["model"]["action_activate"], cfg["model"]["clipAction"])
                loss = (tea_act - stu_act).pow(2).mean()
                grads = list(torch.autograd.grad(loss, [stu[k] for k in names], allow_unused=True))
            else:
                loss, grads = 0.0, None
                for lo, hi in _chunks(len(idx), grad_chunk):
                    w = (hi - lo) / len(idx)
                    mu = net_forward(stu, "actor", cfg["model"]["network"], ring_obs[idx[lo:hi]], cfg.get("proprio_shape", 0))
                    stu_act = action_activation(mu, cfg["model"]["action_activate"], cfg["model"]["clipAction"])
                    loss_c = (tea_act[lo:hi] - stu_act).pow(2).mean()
                    g_c = torch.autograd.grad(loss_c * w, [stu[k] for k in names], allow_unused=True)
                    grads = list(g_c) if grads is None else [a if g is None else a + g for a, g in zip(grads, g_c)]
                    loss = loss + loss_c.detach() * w
            if grad_sync is not None:
                live = [g for g in grads if g is not None]
                grad_sync.mean_grads(live)
                loss = grad_sync.mean_scalar(loss.detach())
            opt.step(grads)
            trace.append(float(loss.detach()))
    lr_now = opt.lrs[0]
    if cfg["lr_schedule"] == "linear_decay":
        lr_now = cfg["lr"] * max(1 - it / cfg["max_iterations"] * 1.8, 0.1)
    elif cfg["lr_schedule"] != "fixed":
        raise NotImplementedError
    opt.lrs = [lr_now for _ in opt.lrs]
    for k in stu:
        stu[k].requires_grad_(False)
    return dict(log={"Train/learning_rate": lr_now, "Train/dagger_loss": sum(trace) / len(trace)},
                loss_trace=trace, opt=opt)


# =============================================================================
# RMS.py (rollout side, "next" row)
# =============================================================================
class RunningMeanStd:
    """RMS.py:3-34."""

    def __init__(self, shape):
        self.n = 0
        self.mean = torch.zeros((1, shape))
        self.S = torch.ones((1, shape)) * 1e-4
        self.std = torch.sqrt(self.S)

    def update(self, x):
        self.n += 1
        old = self.mean.clone()
        new = x.mean(dim=0, keepdim=True)
        self.mean = old + (new - old) / self.n
        self.S = self.S + (x - new).pow(2).mean(dim=0, keepdim=True) + (old - new).pow(2) * (self.n - 1) / self.n
        self.std = torch.sqrt(self.S / self.n)


    def normalize(self, x, update=True):
        """RMS.py:40-45 `Normalization.__call__`."""
        if update:
            self.update(x)
        return (x - self.mean) / self.std


def random_act_cri(p, model_cfg, obs, eps, proprio_shape=0):
    """actor_critic.py:36-47: `MultivariateNormal(mu, scale_tril=diag(exp(log_std)^2)).sample()` = mu + sigma^2 * eps
    with the standard-normal draw `eps` passed in; returns (squashed actions, log_prob of the raw sample, value, mu,
    log_std rows)."""
    mu = net_forward(p, "actor", model_cfg["network"], obs, proprio_shape)
    sig2 = p["log_std"].exp() * p["log_std"].exp()
    x = mu + sig2 * eps
    logp, _ = gaussian_logp_entropy(mu, p["log_std"], x)
    value = net_forward(p, "critic", model_cfg["network"], obs, proprio_shape)
    return (action_activation(x, model_cfg["action_activate"], model_cfg["clipAction"]), logp, value, mu,
            p["log_std"].repeat(mu.shape[0], 1))


# =============================================================================
# Point-set operators -- PARITY UNPINNED (no implementation in the reference tree)
# =============================================================================
def fps(points, K, lengths=None):
    """Farthest point sampling, pytorch3d `sample_farthest_points` defaults as called at
    utils/depth2tsdf.py:113,160: start index 0, squared-L2 running-min distance, next =
    argmax with lowest-index tie-break.  points (B,P,D) float32 -> idx (B,K) int64.
    If K > length the tail is padded with -1 (pytorch3d semantics)."""
    pts = np.asarray(points, dtype=np.float32)
    B, P, _ = pts.shape
    out = np.full((B, K), -1, dtype=np.int64)
    for b in range(B):
        n = P if lengths is None else int(lengths[b])
        if n == 0:
            continue
        mind = np.full((n,), np.inf, dtype=np.float32)
        sel = 0
        for j in range(min(K, n)):
            out[b, j] = sel
            d = pts[b, :n] - pts[b, sel]
            d2 = np.zeros((n,), dtype=np.float32)
            for c in range(d.shape[1]):              # fixed left-to-right fp32 accumulation order
                d2 = d2 + d[:, c] * d[:, c]
            mind = np.minimum(mind, d2)
            sel = int(np.argmax(mind))
    return out


def bc_index_batches(n, n_minibatches):
    """bc.py:113-115: `DataLoader(dataset, batch_size=n // n_minibatches, shuffle=True)` -- the index batches of ONE
    epoch, drawn from the global torch RNG exactly as the loader does (base seed, then the sampler's own seed);
    the ragged tail batch is kept (drop_last=False)."""
    loader = torch.utils.data.DataLoader(range(n), batch_size=n // n_minibatches, shuffle=True, num_workers=0)
    return [b.clone() for b in loader]


def bc_run(p, data, cfg, net_cfg):
    """bc.py:109-177 `bc.run()`: per iteration one shuffled pass; loss = mean((action - tanh(mu)*max_a)^2) (bc.py:139),
    Adam over the student's actor parameters (the only ones with a gradient), lr schedule at iteration end."""
    names = [k for k in p if k.startswith("actor.")]
    for k in names:
        p[k].requires_grad_(True)
    opt = Adam([p[k] for k in names], cfg["lr"])
    x_all = torch.cat([data["tsdf"], data["proprio_state"]], dim=-1) if cfg["add_proprio_obs"] else data["tsdf"]
    n = x_all.shape[0]
    losses, lrs, it = [], [], 0
    while it < cfg["max_iterations"]:
        it += 1
        tot, cnt = 0.0, 0
        for idx in bc_index_batches(n, cfg["n_minibatches"]):
            mu = net_forward(p, "actor", net_cfg, x_all[idx], cfg.get("proprio_shape", 0))
            stu_act = action_activation(mu, cfg["action_activate"], cfg["max_action"])
            loss = (data["action"][idx] - stu_act).pow(2).mean()
            grads = torch.autograd.grad(loss, [p[k] for k in names])
            opt.step(list(grads))
            tot += float(loss.detach())
            cnt += 1
        lr_now = opt.lrs[0]
        if cfg["lr_schedule"] == "linear_decay":
            lr_now = cfg["lr"] * (1 - it / cfg["max_iterations"])
        elif cfg["lr_schedule"] == "step_decay":
            lr_now = cfg["lr"] if it < cfg["max_iterations"] / 2 else cfg["lr"] * 0.1
        elif cfg["lr_schedule"] != "fixed":
            raise NotImplementedError
        opt.lrs = [lr_now for _ in opt.lrs]
        losses.append(tot / cnt)
        lrs.append(lr_now)
    return dict(loss_trace=losses, lr_trace=lrs)


def depth2pc(depth, cam_pose, cam_intr, size, vol_origin, K=1024, return_world=False):
    """utils/depth2tsdf.py:136-173 `TSDFVolume.depth2pc`: depth (b,m,h,w) -> (b,K,3).  fp32 op by op in the
    order of the reference's tensor expression -- PINNED bit for bit to the reference's own world cloud
    (fixture generated by tests/golden/make_golden.py) -- then `fps` above in place of
    pytorch3d.ops.sample_farthest_points (absent here: PARITY UNPINNED for the sampling)."""
    f = np.float32
    depth = np.asarray(depth, dtype=f)
    b, m, h, w = depth.shape
    pose = np.asarray(cam_pose, dtype=f)
    cx, cy = f(cam_intr[0][2]), f(cam_intr[1][2])
    fx, fy = f(cam_intr[0][0]), f(cam_intr[1][1])
    pt2 = depth.reshape(b, m, h * w)
    xmap = np.repeat(np.arange(h), w).astype(f)                  # row index of each pixel (depth2tsdf.py:64)
    ymap = np.tile(np.arange(w), h).astype(f)                    # column index (depth2tsdf.py:65)
    pt0 = ((ymap - cx) * pt2).astype(f) / fx
    pt1 = ((xmap - cy) * pt2).astype(f) / fy
    R, tr = pose[:, :3, :3], pose[:, :3, 3]
    world = np.empty((b, m, h * w, 3), dtype=f)
    def fma(a, bb, c):                                           # exact product in fp64, one rounding to fp32
        return (a.astype(np.float64) * bb.astype(np.float64) + c.astype(np.float64)).astype(f)

    for d in range(3):
        # cld @ R^T: torch.bmm's K=3 inner product is the fused chain fma(p2,r2, fma(p1,r1, p0*r0)) -- checked
        # bit for bit against the reference's own output (tests/golden/depth2pc_small.npz); then + t
        r0, r1, r2 = (np.broadcast_to(R[None, :, d, k, None], pt0.shape) for k in range(3))
        acc = fma(pt2, r2, fma(pt1, r1, (pt0 * r0).astype(f)))
        world[..., d] = (acc + tr[None, :, d, None]).astype(f)
    world = world.reshape(b, m * h * w, 3)
    lo = np.asarray(vol_origin, dtype=f)
    hi = (f(size) + lo).astype(f)
    valid = ((world < hi) & (world > lo)).sum(axis=-1, keepdims=True) == 3
    world = (world * valid).astype(f)
    idx = fps(world, K)
    out = np.take_along_axis(world, idx[..., None].repeat(3, axis=-1), axis=1)
    return (out, world, idx) if return_world else out


def tsdf_tables(cam_pose, cam_intr, im_h, im_w, size, resolution, vol_origin):
    """depth2tsdf.py:14-28,41-60 (`__init__` + `register_camera`): per view and voxel the pixel it projects to
    (row*W + col, or -1 outside the frustum) and its camera-frame depth."""
    cam_pose = torch.as_tensor(np.asarray(cam_pose), dtype=torch.float32)
    voxel = size / resolution
    ax = torch.arange(0, resolution)
    xv, yv, zv = torch.meshgrid(ax, ax, ax, indexing="ij")
    vox = torch.stack([xv.flatten(), yv.flatten(), zv.flatten()], dim=1).long()
    world_c = torch.tensor(list(vol_origin), dtype=torch.float32) + (voxel * vox)
    world_c = world_c[None, ...].repeat(cam_pose.shape[0], 1, 1)
    cam_c = torch.bmm(world_c - cam_pose[:, :3, 3].unsqueeze(-2), cam_pose[:, :3, :3])
    fx, fy, cx, cy = float(cam_intr[0][0]), float(cam_intr[1][1]), float(cam_intr[0][2]), float(cam_intr[1][2])
    pix_z = cam_c[..., 2]
    pix_x = torch.round((cam_c[..., 0] * fx / cam_c[..., 2]) + cx).long()
    pix_y = torch.round((cam_c[..., 1] * fy / cam_c[..., 2]) + cy).long()
    valid = (pix_x >= 0) & (pix_x < im_w) & (pix_y >= 0) & (pix_y < im_h) & (pix_z > 0)
    return torch.where(valid, pix_y * im_w + pix_x, torch.full_like(pix_x, -1)), pix_z


def tsdf_integrate(depth, pix_idx, pix_z, size, resolution, default_tsdf=1.0):
    """depth2tsdf.py:68-86 `TSDFVolume.integrate`: depth (b,m,h,w) -> (b, res, res, res)."""
    depth = torch.as_tensor(depth, dtype=torch.float32)
    b, m = depth.shape[:2]
    trunc = 4 * (size / resolution)
    valid_pix = pix_idx >= 0
    flat = depth.reshape(b, m, -1)
    depth_val = torch.gather(flat, 2, pix_idx.clamp(min=0).unsqueeze(0).expand(b, -1, -1))
    depth_diff = depth_val - pix_z
    tsdf = torch.clamp(depth_diff / trunc, max=1)
    valid_pts = valid_pix & (depth_val > 0) & (depth_diff >= -trunc)
    n_valid = valid_pts.float().sum(dim=1)
    weight = torch.where(valid_pts != 0, 1.0 / n_valid.unsqueeze(1), torch.zeros(1))
    vol = (tsdf * weight).sum(dim=1) + default_tsdf * (n_valid == 0)
    return vol.reshape(b, resolution, resolution, resolution)


def tsdf_sparse_voxel(vol, K=1024, lo=-0.2, hi=0.2):
    """depth2tsdf.py:103-119 (`TSDFVolume.sparse_voxel` after the integration): per env the voxels with lo < tsdf < hi
    in `torch.where` (row-major) order, farthest point sampling over their integer coordinates (squared distances
    are exact integers, so the float restatement `fps` picks the same voxels), padding rows (fewer candidates than K)
    = voxel (0,0,0) as pytorch3d's zero-filled gather leaves them; returns (b, K, 4) float32 rows (x, y, z, tsdf)."""
    vol = torch.as_tensor(vol, dtype=torch.float32)
    rows = []
    for i in range(vol.shape[0]):
        ind = torch.stack(torch.where((vol[i] < hi) & (vol[i] > lo)), dim=-1)
        idx = torch.from_numpy(fps(ind.unsqueeze(0).numpy().astype(np.float32), K))[0]
        sel = torch.where(idx.unsqueeze(-1) >= 0, ind[idx.clamp(min=0)], torch.zeros(1, dtype=ind.dtype))
        rows.append(sel)
    all_ind = torch.stack(rows).long()
    help_ = torch.arange(all_ind.shape[0]).unsqueeze(-1).repeat(1, all_ind.shape[1])
    val = vol[help_, all_ind[..., 0], all_ind[..., 1], all_ind[..., 2]]
    return torch.cat((all_ind, val.unsqueeze(-1)), dim=-1)


def ball_query(xyz, centers, radius, nsample):
    """PointNet++ ball query: for each centre the first `nsample` point indices (ascending
    index order) with squared distance < radius^2, padded with the first hit; if no point
    qualifies the row is all 0.  xyz (B,P,3), centers (B,S,3) -> idx (B,S,nsample) int32."""
    xyz = np.asarray(xyz, dtype=np.float32)
    ctr = np.asarray(centers, dtype=np.float32)
    B, P, _ = xyz.shape
    S = ctr.shape[1]
    r2 = np.float32(radius) * np.float32(radius)
    out = np.zeros((B, S, nsample), dtype=np.int32)
    for b in range(B):
        for s in range(S):
            d = xyz[b] - ctr[b, s]
            d2 = d[:, 0] * d[:, 0]
            d2 = d2 + d[:, 1] * d[:, 1]
            d2 = d2 + d[:, 2] * d[:, 2]
            hit = np.nonzero(d2 < r2)[0][:nsample]
            if len(hit):
                out[b, s, :] = hit[0]
                out[b, s, :len(hit)] = hit
    return out


def group_points(feat, idx):
    """Gather: feat (B,P,C), idx (B,S,ns) -> (B,S,ns,C)."""
    feat = np.asarray(feat)
    B = feat.shape[0]
    return np.stack([feat[b][idx[b]] for b in range(B)], axis=0)


def fps_torch(points, K):
    """`fps` above for equal-length clouds as batched torch ops on whatever device `points` lives on (the whole-update
    parity tests evaluate the restatement on the GPU through ATen).  Same arithmetic, element for element: fp32
    differences, squares and the left-to-right sum as SEPARATE tensor ops (no fused multiply-add can form across them),
    running minimum, lowest-index arg-max.  tests/test_oracle_pointnet2_rows.py pins it to the loop form."""
    pts = torch.as_tensor(points, dtype=torch.float32)
    B, P, D = pts.shape
    out = torch.full((B, K), -1, dtype=torch.int64, device=pts.device)
    mind = torch.full((B, P), float("inf"), dtype=torch.float32, device=pts.device)
    sel = torch.zeros(B, dtype=torch.int64, device=pts.device)
    ar, lane = torch.arange(B, device=pts.device), torch.arange(P, device=pts.device)
    for j in range(min(K, P)):
        out[:, j] = sel
        d = pts - pts[ar, sel].unsqueeze(1)
        sq = d * d
        d2 = sq[..., 0]
        for c in range(1, D):
            d2 = d2 + sq[..., c]
        mind = torch.minimum(mind, d2)
        top = mind.max(dim=1, keepdim=True)[0]
        sel = torch.where(mind == top, lane, P).min(dim=1)[0]        # lowest index among the maxima
    return out


def ball_query_torch(xyz, centers, radius, nsample, chunk=128):
    """`ball_query` above as batched torch ops on xyz's device: (B,P,3), (B,S,3) -> (B,S,nsample) int32."""
    xyz = torch.as_tensor(xyz, dtype=torch.float32)
    ctr = torch.as_tensor(centers, dtype=torch.float32, device=xyz.device)
    B, P, _ = xyz.shape
    S = ctr.shape[1]
    r2 = float(np.float32(radius) * np.float32(radius))
    lane = torch.arange(P, device=xyz.device)
    out = torch.zeros(B, S, nsample, dtype=torch.int32, device=xyz.device)
    for lo in range(0, B, chunk):
        d = xyz[lo:lo + chunk, None, :, :] - ctr[lo:lo + chunk, :, None, :]
        sq = d * d
        d2 = sq[..., 0] + sq[..., 1]
        d2 = d2 + sq[..., 2]
        key = torch.where(d2 < r2, lane, P)                          # hits keep their index, misses sort behind them
        first = torch.sort(key, dim=-1)[0][..., :nsample]
        if first.shape[-1] < nsample:
            first = F.pad(first, (0, nsample - first.shape[-1]), value=P)
        head = first[..., :1]
        row = torch.where(first < P, first, head)                    # short groups repeat their first hit
        out[lo:lo + chunk] = torch.where(head < P, row, torch.zeros_like(row)).to(torch.int32)      # no hit at all: zeros
    return out


def pointnet2_geometry(x, net_cfg):
    """Centre / neighbour tables of every set-abstraction level for the clouds in x (B, >= P*C): [(idx_c (B,S) int64,
    idx_g (B,S,ns) int64)] -- what `pointnet2_forward` computes inline, as batched torch ops on x's device so that a whole
    rollout's tables are built once (they depend on the coordinates only) and handed to every forward (`geom=`)."""
    P = int(net_cfg.get("point_num", 1024))
    B = x.shape[0]
    C = x.shape[1] // P
    xyz = x[:, :P * C].reshape(B, P, C)[..., :3].to(torch.float32)
    out = []
    for S, r, ns in zip(net_cfg.get("npoints", [256, 64]), net_cfg.get("radii", [0.2, 0.4]), net_cfg.get("nsamples", [32, 32])):
        idx_c = fps_torch(xyz, S)
        centers = torch.gather(xyz, 1, idx_c.unsqueeze(-1).expand(B, S, 3))
        out.append((idx_c, ball_query_torch(xyz, centers, r, ns).long()))
        xyz = centers
    return out


def _pad4(n):
    return (n + 3) // 4 * 4


def pointnet2_forward(p, prefix, net_cfg, x, proprio_shape=0, pool_args=None, return_aux=False, geom=None):
    """PointNet++ single-scale-grouping encoder -- PARITY UNPINNED (absent from the reference; this
    restates the published structure the way partmanip_amd.algo_utils.network.PointNet2 documents it:
    per level FPS -> ball query -> rows [xyz-centre | feat | 0-pad to a multiple of 4] -> shared MLP with
    the activation after every layer -> max over the group; a final group-all level on absolute
    coordinates; the PointNet head).  `pool_args`: optional list of (G, C) index tensors pinning each
    max-pool's arg-max (test hook, see pointnet_forward).  `geom`: the levels' (idx_c, idx_g) tables built ahead by
    `pointnet2_geometry` from the same rows (they depend on the coordinates only); None: sampled / queried here."""
    P = int(net_cfg.get("point_num", 1024))
    B = x.shape[0]
    C = x.shape[1] // P
    act = net_cfg["activation"]
    npoints = list(net_cfg.get("npoints", [256, 64]))
    radii = list(net_cfg.get("radii", [0.2, 0.4]))
    nsamples = list(net_cfg.get("nsamples", [32, 32]))
    n_levels = len(npoints) + 1
    pts = x[:, :P * C].reshape(B, P, C)
    xyz, feat = pts[..., :3], (pts[..., 3:] if C > 3 else None)
    aux, used_args = [], []

    def shared_mlp(l, rows):
        i = 0
        while f"{prefix}.sa.{l}.{2 * i}.weight" in p:
            rows = _act(act, _lin(p, f"{prefix}.sa.{l}.{2 * i}", rows))
            i += 1
        return rows

    def pool(h, G, ns, k):
        h = h.reshape(G, ns, -1)
        if pool_args is not None:
            idx = pool_args[k].long()
            out = torch.gather(h, 1, idx.unsqueeze(1)).squeeze(1)
        else:
            out, idx = h.max(dim=1)
        used_args.append(idx)
        return out

    for l in range(n_levels - 1):
        S, ns = npoints[l], nsamples[l]
        if geom is not None:
            idx_c, idx_g = geom[l]
            centers = torch.gather(xyz, 1, idx_c.unsqueeze(-1).expand(B, S, 3))
        else:
            on_host = x.device.type == "cpu"                                         # (elsewhere: the batched torch forms)
            idx_c = torch.from_numpy(fps(xyz.detach().numpy(), S)) if on_host else fps_torch(xyz.detach(), S)      # (B,S)
            centers = torch.gather(xyz, 1, idx_c.unsqueeze(-1).expand(B, S, 3))
            idx_g = torch.from_numpy(ball_query(xyz.detach().numpy(), centers.detach().numpy(), radii[l], ns)).long() if on_host \
                else ball_query_torch(xyz.detach(), centers.detach(), radii[l], ns).long()
        # rows of the cloud's points picked by index (index_select on the flattened batch: the same rows as a per-cloud gather,
        # with an index_add as its backward instead of a scatter through an expanded index)
        Pl = xyz.shape[1]
        flat = (idx_g.reshape(B, S * ns) + (torch.arange(B, device=x.device) * Pl).view(B, 1)).reshape(-1)
        g_xyz = xyz.reshape(B * Pl, 3).index_select(0, flat).reshape(B, S, ns, 3) - centers.unsqueeze(2)
        cols = [g_xyz]
        cf = 0
        if feat is not None:
            cf = feat.shape[2]
            cols.append(feat.reshape(B * Pl, cf).index_select(0, flat).reshape(B, S, ns, cf))
        pad = _pad4(3 + cf) - (3 + cf)
        if pad:
            cols.append(torch.zeros(B, S, ns, pad, dtype=x.dtype, device=x.device))
        rows = torch.cat(cols, dim=-1).reshape(B * S * ns, -1)
        pooled = pool(shared_mlp(l, rows), B * S, ns, l)
        aux.append((idx_c, idx_g))
        xyz, feat = centers, pooled.reshape(B, S, -1)
    S, cf = xyz.shape[1], feat.shape[2]
    cols = [xyz, feat]
    pad = _pad4(3 + cf) - (3 + cf)
    if pad:
        cols.append(torch.zeros(B, S, pad, dtype=x.dtype, device=x.device))
    rows = torch.cat(cols, dim=-1).reshape(B * S, -1)
    f = pool(shared_mlp(n_levels - 1, rows), B, S, n_levels - 1)
    if proprio_shape != 0:
        f = torch.cat((f, x[:, -proprio_shape:]), dim=-1)
    f = _act(act, _lin(p, f"{prefix}.final_mlp.0", f))
    f = _act(act, _lin(p, f"{prefix}.final_mlp.2", f))
    out = _lin(p, f"{prefix}.final_mlp.4", f)
    return (out, aux, used_args) if return_aux else out


# =============================================================================
# 3D sparse-voxel U-Net -- PARITY UNPINNED (README.md:30 names it, README.md:23: the code is not in the snapshot)
# =============================================================================
def sparse_unet_geometry(x, P, C, R):
    """Index tables of partmanip_amd.algo_utils.network.SparseUNet for the clouds in x (B, >= P*C): numpy restatement of
    csrc/sparse_voxel.hip.  Rows of a level are numbered cloud by cloud; level 0 in input order, levels 1-2 in cell order
    ((X*Rc + Y)*Rc + Z); a duplicate coordinate resolves to its lowest row."""
    xs = np.asarray(x, dtype=np.float32)
    B = xs.shape[0]
    pts = xs[:, :P * C].reshape(B, P, C)
    co = np.clip(np.floor(pts[..., :3]).astype(np.int64), 0, R - 1)
    f = pts[..., 3] if C > 3 else np.ones((B, P), np.float32)
    feat0 = np.concatenate([f[..., None], co.astype(np.float32) / np.float32(R)], axis=-1).reshape(B * P, 4).astype(np.float32)
    offs = [(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]

    def level_tables(coords, Rl):
        """coords: list per cloud of (n, 3) int arrays (row order) -> (canonical map per cloud, nbr (rows, 27))"""
        maps, nbr, base = [], [], 0
        for cb in coords:
            m = {}
            for i, c in enumerate(map(tuple, cb)):
                m.setdefault(c, base + i)
            maps.append(m)
            for c in cb:
                row = []
                for d in offs:
                    q = (c[0] + d[0], c[1] + d[1], c[2] + d[2])
                    ok = all(0 <= q[k] < Rl for k in range(3))
                    row.append(m.get(q, -1) if ok else -1)
                nbr.append(row)
            base += len(cb)
        return maps, np.asarray(nbr, dtype=np.int64).reshape(-1, 27)

    def down(coords, maps, Rf):
        Rc = (Rf + 1) // 2
        cc, child, parent, parent_canon, slot = [], [], [], [], []
        base_c, base_f = 0, 0
        for cb, m in zip(coords, maps):
            cells = sorted({(c[0] >> 1, c[1] >> 1, c[2] >> 1) for c in map(tuple, cb)}, key=lambda q: (q[0] * Rc + q[1]) * Rc + q[2])
            pm = {q: base_c + i for i, q in enumerate(cells)}
            for q in cells:
                child.append([m.get((2 * q[0] + (s >> 2), 2 * q[1] + ((s >> 1) & 1), 2 * q[2] + (s & 1)), -1) for s in range(8)])
            for i, c in enumerate(map(tuple, cb)):
                p_ = pm[(c[0] >> 1, c[1] >> 1, c[2] >> 1)]
                parent.append(p_)
                parent_canon.append(p_ if m[c] == base_f + i else -1)
                slot.append((c[0] & 1) * 4 + (c[1] & 1) * 2 + (c[2] & 1))
            cc.append(np.asarray(cells, dtype=np.int64).reshape(-1, 3))
            base_c += len(cells)
            base_f += len(cb)
        i64 = lambda a: np.asarray(a, dtype=np.int64)
        return dict(R=Rc, coords=cc, rows=base_c, child=i64(child).reshape(-1, 8), parent=i64(parent), parent_canon=i64(parent_canon),
                    slot=i64(slot))

    c0 = [co[b] for b in range(B)]
    m0, nbr0 = level_tables(c0, R)
    l1 = down(c0, m0, R)
    m1, nbr1 = level_tables(l1["coords"], l1["R"])
    l2 = down(l1["coords"], m1, l1["R"])
    _, nbr2 = level_tables(l2["coords"], l2["R"])
    return dict(feat0=feat0, nbr0=nbr0, nbr1=nbr1, nbr2=nbr2, l1=l1, l2=l2, rows=(B * P, l1["rows"], l2["rows"]))


def sparse_unet_geometry_torch(x, P, C, R):
    """`sparse_unet_geometry` above as batched torch ops on x's device (dense per-cloud index grids instead of dicts): the same
    tables as int64 tensors, level coordinates as one (rows, 3) tensor + the rows' cloud ids instead of per-cloud lists.
    tests/test_oracle_sparse_unet.py pins it to the dict form."""
    dev = x.device
    B = x.shape[0]
    pts = x[:, :P * C].reshape(B, P, C).to(torch.float32)
    co = torch.floor(pts[..., :3]).long().clamp(0, R - 1)
    f = pts[..., 3] if C > 3 else torch.ones(B, P, device=dev)
    feat0 = torch.cat([f.unsqueeze(-1), co.to(torch.float32) / torch.full((), R, dtype=torch.float32, device=dev)], dim=-1).reshape(B * P, 4)
    offs = torch.tensor([(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)], device=dev)
    BIG = torch.iinfo(torch.int64).max

    def canon_grid(coords, cloud, Rl):
        """flat (B * Rl^3) grid: lowest row with that coordinate, -1 where the cell is empty"""
        key = cloud * Rl ** 3 + (coords[:, 0] * Rl + coords[:, 1]) * Rl + coords[:, 2]
        grid = torch.full((B * Rl ** 3,), BIG, dtype=torch.int64, device=dev)
        grid.scatter_reduce_(0, key, torch.arange(coords.shape[0], device=dev), reduce="amin")
        return torch.where(grid == BIG, torch.full_like(grid, -1), grid), key

    def lookup(grid, q, cloud, Rl):
        ok = ((q >= 0) & (q < Rl)).all(dim=-1)
        qc = q.clamp(0, Rl - 1)
        key = cloud * Rl ** 3 + (qc[..., 0] * Rl + qc[..., 1]) * Rl + qc[..., 2]
        return torch.where(ok, grid[key], torch.full_like(key, -1))

    def down(coords, cloud, grid, key, Rf):
        Rc = (Rf + 1) // 2
        cell = coords >> 1
        ck = cloud * Rc ** 3 + (cell[:, 0] * Rc + cell[:, 1]) * Rc + cell[:, 2]
        uniq, parent = torch.unique(ck, sorted=True, return_inverse=True)         # coarse rows: cloud by cloud, cell order
        ccloud, rem = uniq // Rc ** 3, uniq % Rc ** 3
        cc = torch.stack([rem // (Rc * Rc), (rem // Rc) % Rc, rem % Rc], dim=1)
        bits = torch.tensor([(s >> 2, (s >> 1) & 1, s & 1) for s in range(8)], device=dev)
        child = lookup(grid, 2 * cc[:, None, :] + bits[None], ccloud[:, None], Rf)
        is_canon = grid[key] == torch.arange(coords.shape[0], device=dev)
        slot = (coords[:, 0] & 1) * 4 + (coords[:, 1] & 1) * 2 + (coords[:, 2] & 1)
        return dict(R=Rc, coords=cc, cloud=ccloud, rows=int(uniq.numel()), child=child, parent=parent,
                    parent_canon=torch.where(is_canon, parent, torch.full_like(parent, -1)), slot=slot)

    c0 = co.reshape(B * P, 3)
    b0 = torch.arange(B, device=dev).repeat_interleave(P)
    g0, k0 = canon_grid(c0, b0, R)
    nbr0 = lookup(g0, c0[:, None, :] + offs[None], b0[:, None], R)
    l1 = down(c0, b0, g0, k0, R)
    g1, k1 = canon_grid(l1["coords"], l1["cloud"], l1["R"])
    nbr1 = lookup(g1, l1["coords"][:, None, :] + offs[None], l1["cloud"][:, None], l1["R"])
    l2 = down(l1["coords"], l1["cloud"], g1, k1, l1["R"])
    g2, _ = canon_grid(l2["coords"], l2["cloud"], l2["R"])
    nbr2 = lookup(g2, l2["coords"][:, None, :] + offs[None], l2["cloud"][:, None], l2["R"])
    return dict(feat0=feat0, nbr0=nbr0, nbr1=nbr1, nbr2=nbr2, l1=l1, l2=l2, rows=(B * P, l1["rows"], l2["rows"]))


def _rows_gather(src, idx):
    """(rows, J) index table -> (rows, J*C): neighbour rows side by side, zeros where idx < 0."""
    idx = torch.as_tensor(idx, device=src.device)
    if idx.dim() == 1:
        idx = idx.view(-1, 1)
    # (index_select: the same rows as src[idx]; its backward is an index_add instead of a sort over every index)
    g = src.index_select(0, idx.clamp(min=0).reshape(-1)).view(idx.shape[0], idx.shape[1], -1) * (idx >= 0).unsqueeze(-1).to(src.dtype)
    return g.reshape(idx.shape[0], -1)


def sparse_unet_forward(p, prefix, net_cfg, x, proprio_shape=0, return_aux=False):
    """The SparseUNet backbone (see the class docstring in partmanip_amd/algo_utils/network.py) in plain torch."""
    P = int(net_cfg.get("point_num", 1024))
    R = int(net_cfg.get("grid", 50))
    B = x.shape[0]
    C = x.shape[1] // P
    act = net_cfg["activation"]
    # (a tensor that lives on a device: the batched torch form of the same tables -- the dict form is host Python)
    g = sparse_unet_geometry(x.detach().numpy(), P, C, R) if x.device.type == "cpu" else sparse_unet_geometry_torch(x.detach(), P, C, R)
    lin = lambda n, v: _act(act, _lin(p, f"{prefix}.{n}", v))
    F0 = torch.as_tensor(g["feat0"]).to(x.dtype)
    H0 = lin("conv0", _rows_gather(F0, g["nbr0"]))
    D1 = lin("down0", _rows_gather(H0, g["l1"]["child"]))
    H1 = lin("conv1", _rows_gather(D1, g["nbr1"]))
    D2 = lin("down1", _rows_gather(H1, g["l2"]["child"]))
    H2 = lin("conv2", _rows_gather(D2, g["nbr2"]))
    E1 = lin("up1", torch.cat([_rows_gather(H2, g["l2"]["parent"]), H1], dim=1))
    E0 = lin("up0", torch.cat([_rows_gather(E1, g["l1"]["parent"]), H0], dim=1))
    feat = E0.view(B, P, -1).max(dim=1)[0]
    if proprio_shape:
        feat = torch.cat([feat, x[:, -proprio_shape:]], dim=1)
    h = _act(act, _lin(p, f"{prefix}.final_mlp.0", feat))
    h = _act(act, _lin(p, f"{prefix}.final_mlp.2", h))
    out = _lin(p, f"{prefix}.final_mlp.4", h)
    return (out, dict(g=g, E0=E0)) if return_aux else out
