"""torch.autograd bridge over the HIP forward / backward entry points ("Level 1.5" of INTEGRATION.md).

The learners in `partmanip_amd.algorithms` never build an autograd graph: they call the explicit HIP backward.  The
REFERENCE's own `update()` does -- it calls `loss.backward()` on what `ActorCritic.update_act_cri` returns
(ppo.py:326,347-348; dagger.py:312-318; actor_critic.py:71-82) and steps `torch.optim.Adam`.  With
`ActorCritic.autograd = True` the same methods return graph-carrying tensors whose backward IS the HIP backward, so
a reference-shaped training loop (its own `update()`, `clip_grad_norm_`, `torch.optim.Adam`) drives the MI355X
backbones unchanged.  torch only routes gradients here; every derivative is computed by a kernel behind the C ABI.
"""
import torch

from . import ops


class BackboneFn(torch.autograd.Function):
    """y = net(x) for any backbone with `hip_forward` / `hip_backward` (MLP, PointNet, PointNet2, Conv3DNet).
    `params` are listed only so that autograd routes the parameter gradients; the kernels read the parameters'
    storage directly and write the gradients into the network's gradient views (ActorCritic.flatten)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        ctx.net = net
        ctx.n_params = len(params)
        y = net.hip_forward(x)
        net._fwd_serial = getattr(net, "_fwd_serial", 0) + 1          # the backbone keeps ONE set of saved activations
        ctx.serial = net._fwd_serial
        return y

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        if getattr(net, "_fwd_serial", 0) != ctx.serial:
            raise RuntimeError("BackboneFn.backward: the backbone ran another training forward since this graph was built "
                               "(one set of saved activations per backbone: call backward before the next forward)")
        views = getattr(net, "_grad_list", None)
        if views is None or len(views) != ctx.n_params:
            raise RuntimeError("BackboneFn needs the owner's flat gradient views (ActorCritic.flat())")
        net.hip_backward(dy.contiguous())
        return (None, None) + tuple(v.clone() for v in views)


def backbone_apply(net, x):
    return BackboneFn.apply(net, x, *net.parameters())


class GaussianLogpFn(torch.autograd.Function):
    """(log_prob (B,), entropy (B,)) of actor_critic.py:74-78 for squashed actions, differentiable in mu and log_std."""

    @staticmethod
    def forward(ctx, mu, log_std, actions, max_action, act_tanh):
        B = mu.shape[0]
        mu, actions = mu.contiguous(), actions.contiguous()
        logp = torch.empty(B, device=mu.device)
        ent = torch.empty(B, device=mu.device)
        ops.gaussian_logp(mu, log_std, actions, max_action, act_tanh, logp, ent)
        ctx.save_for_backward(mu, log_std, actions)
        ctx.cfg = (max_action, act_tanh)
        return logp, ent

    @staticmethod
    def backward(ctx, dlogp, dent):
        mu, log_std, actions = ctx.saved_tensors
        dmu = torch.empty_like(mu)
        dls = torch.empty_like(log_std)
        ops.gaussian_logp_bwd(mu, log_std, actions, ctx.cfg[0], ctx.cfg[1], None if dlogp is None else dlogp.contiguous(),
                              None if dent is None else dent.contiguous(), dmu, dls)
        return dmu, dls, None, None, None


class ActionActivationFn(torch.autograd.Function):
    """actor_critic.py:84-91: tanh(mu) * max_action."""

    @staticmethod
    def forward(ctx, mu, max_action):
        mu = mu.contiguous()
        out = torch.empty_like(mu)
        ops.action_activation(mu, out, max_action, True)
        ctx.save_for_backward(out)
        ctx.max_action = max_action
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dmu = torch.empty_like(out)
        ops.action_activation_bwd(out, dout.contiguous(), dmu, ctx.max_action, True)
        return dmu, None
