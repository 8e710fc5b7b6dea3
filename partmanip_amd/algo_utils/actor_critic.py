"""ActorCritic with the reference's interface (algorithms/algo_utils/actor_critic.py:8-100):
two separate backbones + a `log_std` parameter, a Gaussian policy whose covariance factor is
diag(exp(log_std)^2) (so the effective std is sigma^2 -- kept deliberately),
tanh action squashing.  Same method names, argument meaning and return tuples.

Added for the MI355X path: `flatten()` re-homes all parameters into two contiguous fp32
buffers -- [actor params | log_std] and [critic params] -- matching the reference's two
optimisers (ppo.py:73-74), with same-layout gradient buffers the HIP backward writes into.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .network import MLP, Conv3DNet, PoolConv3DNet, PointNet, PointNet2, SparseUNet, ResNet, depthResNet  # noqa: F401 (eval() namespace)
from .. import ops


class ActorCritic(nn.Module):
    SCAL_TAIL = 8
    # split-K slabs of the grouped weight gradients of the small-step path (slab 0 = the gradient buffer itself, the optimiser's
    # norm pass sums the rest).  cfg 2, all four weight gradients of a network in one LDS-DMA launch: 2 slabs 2.13 M env-steps/s,
    # 3 -> 2.26 M, 4 -> 2.23 M, 5 -> 2.19 M, 8 -> 2.20 M (while the unaligned first-layer / head problems still ran as a launch of
    # their own on the register-staged kernel, 8 was best: 1.89 M against 1.85 M with 4).
    GRAD_SLABS = int(os.environ.get("PARTMANIP_GRAD_SLABS", "3"))

    def __init__(self, obs_shape, actions_shape, model_cfg, proprio_shape=0):
        super(ActorCritic, self).__init__()
        net_cfg = model_cfg['network']
        self.actor = eval(net_cfg['name'])(obs_shape, actions_shape, net_cfg, proprio_shape=proprio_shape)
        self.critic = eval(net_cfg['name'])(obs_shape, 1, net_cfg, proprio_shape=proprio_shape)
        self.log_std = nn.Parameter(np.log(model_cfg['action_std']) * torch.ones(actions_shape))
        self.max_action = model_cfg['clipAction']
        assert self.max_action > 0
        self.action_activate = model_cfg['action_activate']
        if self.action_activate not in ('tanh', None):
            raise NotImplementedError
        self.actions_shape = actions_shape
        self._flat = None
        # True: update_act_cri / update_act return tensors that carry a torch autograd graph whose backward is the HIP
        # backward (partmanip_amd/autograd.py) -- for callers that differentiate through them like the reference's own
        # update() (ppo.py:326,347-348).  The runners of this package leave it off and call the HIP backward directly.
        self.autograd = False

    # ------------------------------------------------------------------ flat buffers
    def flatten(self):
        """(Re)build the flat parameter / gradient buffers on the parameters' current device."""
        dev = self.log_std.device
        a_params = list(self.actor.named_parameters())
        c_params = list(self.critic.named_parameters())
        n_a = sum(p.numel() for _, p in a_params)
        n_c = sum(p.numel() for _, p in c_params)
        A = self.log_std.numel()
        flat_a = torch.empty(n_a + A, device=dev)
        flat_c = torch.empty(n_c, device=dev)
        # gradient buffers carry an 8-float tail for the step's scalars (loss, kl, skip flag, ...):
        # a data-parallel step all-reduces gradient and scalars as ONE message (dist.py)
        # ... and is slab 0 of GRAD_SLABS equally spaced slabs: the grouped weight-gradient launch of the small-step path
        # writes split-K partial sums to slabs 0..S-1 and the optimiser launch adds them up (no separate reduce kernels)
        stride_a = (n_a + A + self.SCAL_TAIL + 3) // 4 * 4
        stride_c = (n_c + self.SCAL_TAIL + 3) // 4 * 4
        slabs_a = torch.zeros(self.GRAD_SLABS * stride_a, device=dev)
        slabs_c = torch.zeros(self.GRAD_SLABS * stride_c, device=dev)
        grad_a = slabs_a[:n_a + A + self.SCAL_TAIL]
        grad_c = slabs_c[:n_c + self.SCAL_TAIL]

        def rehome(params, flat, grad):
            views, off = {}, 0
            for name, p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                views[name] = grad[off:off + n].view(p.shape)
                off += n
            return views, off
        va, off = rehome(a_params, flat_a, grad_a)
        flat_a[off:off + A].copy_(self.log_std.data)
        self.log_std.data = flat_a[off:off + A]
        vc, _ = rehome(c_params, flat_c, grad_c)
        self.actor.set_grad_views(va)
        self.critic.set_grad_views(vc)
        # each network's slice of the flat parameter buffer (parameters() order): the source of its operand-copy gather
        object.__setattr__(self.actor, "_param_flat", flat_a[:n_a])
        object.__setattr__(self.critic, "_param_flat", flat_c[:n_c])
        object.__setattr__(self.actor, "_grad_list", [va[n] for n, _ in a_params])     # parameter order (autograd bridge)
        object.__setattr__(self.critic, "_grad_list", [vc[n] for n, _ in c_params])
        self._flat = dict(actor=flat_a, critic=flat_c, grad_actor=grad_a, grad_critic=grad_c, n_actor=n_a,
                          n_critic=n_c, grad_log_std=grad_a[n_a:n_a + A], scal_actor=grad_a[n_a + A:],
                          scal_critic=grad_c[n_c:], slab_stride_actor=stride_a, slab_stride_critic=stride_c,
                          extra_actor=slabs_a[stride_a:], extra_critic=slabs_c[stride_c:])
        return self._flat

    def flat(self):
        f = self._flat
        if f is None or f["actor"].device != self.log_std.device or \
                self.log_std.data_ptr() != f["actor"].data_ptr() + 4 * f["n_actor"] or \
                next(self.critic.parameters()).data_ptr() != f["critic"].data_ptr():
            f = self.flatten()      # first use, or .to(device)/load re-created the parameter tensors
        return f

    def forward(self):
        raise NotImplementedError

    # ------------------------------------------------------------------ reference API
    def _sigma2(self):
        return self.log_std.data.exp() * self.log_std.data.exp()

    def _logp_entropy(self, mu, actions_raw_or_squashed, squashed):
        B = mu.shape[0]
        logp = torch.empty(B, device=mu.device)
        ent = torch.empty(B, device=mu.device)
        ops.gaussian_logp(mu, self.log_std.data, actions_raw_or_squashed.contiguous(), self.max_action,
                          squashed and self.action_activate == 'tanh', logp, ent)
        return logp, ent

    def cri(self, observations):
        self.flat()
        return self.critic(observations).detach()

    def random_act_cri(self, observations):
        """actor_critic.py:36-47."""
        self.flat()
        mu = self.actor(observations)
        # MultivariateNormal(loc, scale_tril=diag(sigma^2)).sample(): loc + sigma^2 * eps
        eps = torch.normal(torch.zeros_like(mu), torch.ones_like(mu))
        actions, logp, log_std_rows = ops.gaussian_sample(mu, self.log_std.data, eps, self.max_action,
                                                          self.action_activate == 'tanh')
        value = self.critic(observations)
        return actions, logp, value, mu, log_std_rows

    def random_act(self, observations):
        self.flat()
        mu = self.actor(observations)
        eps = torch.normal(torch.zeros_like(mu), torch.ones_like(mu))
        return ops.gaussian_sample(mu, self.log_std.data, eps, self.max_action, self.action_activate == 'tanh',
                                   want_log_std_rows=False)[0]

    def act(self, observations):
        self.flat()
        return self.action_activation(self.actor(observations))

    def act_cri(self, observations):
        self.flat()
        return self.action_activation(self.actor(observations)), self.critic(observations)

    def update_act(self, observations):
        self.flat()
        if self.autograd and torch.is_grad_enabled():                    # dagger.py:312: graph back to the student actor
            from ..autograd import backbone_apply, ActionActivationFn
            mu = backbone_apply(self.actor, observations)
            return ActionActivationFn.apply(mu, self.max_action) if self.action_activate == 'tanh' else mu
        return self.action_activation(self.actor(observations))

    def update_act_cri(self, observations, actions):
        """actor_critic.py:71-82 -> (log_prob (B,), entropy (B,), value (B,1), mu (B,A), log_std rows (B,A)).
        Default: values only (the runners of this package differentiate through the explicit HIP backward in
        `algorithms/ppo.py`).  With `self.autograd = True` the five tensors carry an autograd graph to the parameters
        whose backward is that same HIP backward -- `loss.backward()` + `torch.optim.Adam`, as ppo.py:347-353, work."""
        self.flat()
        if self.autograd and torch.is_grad_enabled():
            from ..autograd import backbone_apply, GaussianLogpFn
            mu = backbone_apply(self.actor, observations)
            logp, ent = GaussianLogpFn.apply(mu, self.log_std, actions, self.max_action, self.action_activate == 'tanh')
            value = backbone_apply(self.critic, observations)
            return logp, ent, value, mu, self.log_std.repeat(mu.shape[0], 1)
        mu = self.actor(observations)
        logp, ent = self._logp_entropy(mu, actions, squashed=True)
        value = self.critic(observations)
        return logp, ent, value, mu, self.log_std.data.repeat(mu.shape[0], 1)

    def action_activation(self, action):
        if self.action_activate == 'tanh':
            out = torch.empty_like(action)
            ops.action_activation(action.contiguous(), out, self.max_action, True)
            return out
        elif self.action_activate is None:
            return action
        raise NotImplementedError

    def action_deactivation(self, action):
        if self.action_activate == 'tanh':
            return torch.atanh(torch.clamp(action / self.max_action, max=1 - 1e-5, min=-1 + 1e-5))
        elif self.action_activate is None:
            return action
        raise NotImplementedError
