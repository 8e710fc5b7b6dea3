"""FusedAdam: torch.optim.Adam's interface (param_groups / state_dict / load_state_dict, as
ppo.py:73-74,90-91,116-117,392-400 and dagger.py:56,87,113 use it) over ONE flat fp32 buffer,
stepped by a single fused clip+Adam HIP launch pair (K10).  The step counter and the
"skip this mini-batch" predicate (ppo.py:337-338) live on the device.
"""
import torch

from .. import ops


class FusedAdam:
    def __init__(self, flat_params, flat_grads, groups, lr, betas=(0.9, 0.999), eps=1e-8):
        """flat_params/flat_grads: 1-D fp32 device tensors of equal length.
        groups: list of lists of nn.Parameter (views into flat_params, in buffer order) -- kept so that
        `param_groups` / `state_dict()` look exactly like torch.optim.Adam's."""
        self.p, self.g = flat_params, flat_grads
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)
        self.state_dev = torch.zeros(4, dtype=torch.int32, device=flat_params.device)
        self.gnorm = torch.zeros(1, device=flat_params.device)
        self.ws = ops.Workspace(flat_params.device)
        self.param_groups = [dict(params=list(g), lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
                             for g in groups]
        self._touched = set()          # group indices that have received a gradient (for state_dict)

    def rebind(self, flat_params, flat_grads):
        """Follow the owner's flat buffers after `ActorCritic.flatten()` rebuilt them (a later `.to()` / `.float()` /
        `load_state_dict(assign=True)` re-creates the parameter tensors): stepping the orphaned old buffers would
        silently stop training.  The Adam moments and the step counter carry over."""
        if flat_params.numel() != self.p.numel() or flat_grads.numel() != self.g.numel():
            raise RuntimeError("FusedAdam.rebind: the parameter set changed size")
        dev = flat_params.device
        self.p, self.g = flat_params, flat_grads
        self.m, self.v, self.state_dev, self.gnorm = (t.to(dev) for t in (self.m, self.v, self.state_dev, self.gnorm))
        if self.ws.device != dev:
            self.ws = ops.Workspace(dev)

    def bound_to(self, flat_params):
        return self.p.data_ptr() == flat_params.data_ptr()

    def zero_grad(self, set_to_none=False):
        self.g.zero_()

    def step(self, n=None, n_clip=0, max_norm=0.0, skip_flag=None):
        """One Adam step over the first `n` elements (default all).  The L2-norm clip covers the
        first `n_clip` elements.  All param groups share one lr (the reference always sets them
        together, ppo.py:392-400)."""
        g0 = self.param_groups[0]
        n = self.p.numel() if n is None else n
        ops.clip_adam_step(self.p[:n], self.g[:n], self.m[:n], self.v[:n], n_clip, max_norm, g0['lr'], g0['betas'][0],
                           g0['betas'][1], g0['eps'], self.state_dev, skip_flag, self.gnorm, self.ws)

    def group_item(self, n=None, n_clip=0, max_norm=0.0, skip_flag=None, extra=None, extra_stride=0, n_sum=0, n_extra=0, stats=None,
                   dp=None):
        """The arguments of `step` as one record of ops.clip_adam_group (several optimisers per launch pair; `extra`: the
        split-K gradient slabs 1.. that the norm pass adds onto g[:n_sum] first; `dp` = (1 / world, scalar record or None,
        desired_kl or 0): the data-parallel form after an all-reduce SUM, pm_clip_adam_desc in include/partmanip_hip.h)."""
        g0 = self.param_groups[0]
        n = self.p.numel() if n is None else n
        return dict(p=self.p[:n], g=self.g[:n], m=self.m[:n], v=self.v[:n], n_clip=n_clip, max_norm=max_norm, lr=g0['lr'],
                    b1=g0['betas'][0], b2=g0['betas'][1], eps=g0['eps'], state=self.state_dev, skip_flag=skip_flag,
                    gnorm=self.gnorm, ws=self.ws, extra=extra, extra_stride=extra_stride, n_sum=n_sum, n_extra=n_extra, stats=stats,
                    dp=dp)

    # ---- torch.optim.Adam-compatible (de)serialisation ---------------------------------------
    def _param_slices(self):
        """(offset into the flat buffer | None, numel, shape) per parameter in torch's index order.
        Offsets come from pointer arithmetic, so the groups may list parameters in any order
        (DAgger's Adam enumerates log_std first, dagger.py:56) and may include parameters that
        live outside this buffer (they never receive gradients and never own state)."""
        base, total, out = self.p.data_ptr(), self.p.numel(), []
        for g in self.param_groups:
            for p in g['params']:
                d = p.data_ptr() - base
                off = d // 4 if (d >= 0 and d % 4 == 0 and d // 4 + p.numel() <= total) else None
                out.append((off, p.numel(), p.shape))
        return out

    def state_dict(self, active=None):
        """`active`: optional set of parameter indices that own optimiser state (those that have
        received gradients); default: every parameter inside the flat buffer."""
        step = float(self.state_dev[0].item())
        sl = self._param_slices()
        state = {}
        for i, (off, n, shape) in enumerate(sl):
            if active is not None and i not in active:
                continue
            if step > 0 and off is not None:
                state[i] = dict(step=torch.tensor(step), exp_avg=self.m[off:off + n].view(shape).clone(),
                                exp_avg_sq=self.v[off:off + n].view(shape).clone())
        groups, k = [], 0
        for g in self.param_groups:
            d = {key: val for key, val in g.items() if key != 'params'}
            d['params'] = list(range(k, k + len(g['params'])))
            k += len(g['params'])
            groups.append(d)
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        sl = self._param_slices()
        step = 0
        for i, st in sd['state'].items():
            off, n, _ = sl[int(i)]
            if off is None:
                continue
            self.m[off:off + n].copy_(st['exp_avg'].reshape(-1))
            self.v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
            step = int(float(st['step']))
        self.state_dev[0] = step
        for g, gs in zip(self.param_groups, sd['param_groups']):
            for key in ('lr', 'betas', 'eps'):
                g[key] = gs[key]
