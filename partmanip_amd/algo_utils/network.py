"""Backbones of the actor-critic, mirroring the reference's plug-in interface
(algorithms/algo_utils/network.py): a backbone is a class `(input_dim, output_dim, net_cfg,
proprio_shape)` with `forward(x: (B,input_dim)) -> (B,output_dim)`, chosen by
`net_cfg['name']` (actor_critic.py:16,19).  Parameter names / shapes equal the reference's
`state_dict()` so checkpoints interchange.

Compute is NOT torch autograd: forward and backward are explicit chains of HIP kernels
(ops.py -> libpartmanip_hip.so) writing into flat parameter / gradient buffers owned by
`ActorCritic` (one contiguous buffer per optimiser => one fused clip+Adam launch and one
RCCL all-reduce per step).
"""
import math

import torch
import torch.nn as nn

from .. import ops

_SUPPORTED_ACT = {"tanh": ops.ACT_TANH}


def get_activation(act_name):
    """network.py:7-24 (module objects only keep the Sequential indices / state_dict keys aligned)."""
    table = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "lrelu": nn.LeakyReLU,
             "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}
    if act_name not in table:
        print("invalid activation function!")
        return None
    return table[act_name]()


def _act_code(name):
    if name not in _SUPPORTED_ACT:
        raise NotImplementedError(f"activation '{name}': the HIP path implements tanh (every shipped cfg); no fallback")
    return _SUPPORTED_ACT[name]


class _LinearChain:
    """Forward/backward of Linear-act-...-Linear over explicit buffers (K4/K5 kernels)."""

    def __init__(self, linears, act_code):
        self.linears = linears            # list[nn.Linear]
        self.act = act_code
        self.h = []                        # saved activation outputs of the hidden layers
        self.grads = None                  # list[(dW view, db view)] set by ActorCritic.flatten()

    def forward(self, x, out=None):
        n = len(self.linears)
        self.x = x
        self.h = []
        cur = x
        for i, lin in enumerate(self.linears):
            last = i == n - 1
            y = out if (last and out is not None) else torch.empty(cur.shape[0], lin.out_features, device=cur.device)
            ops.linear_fwd(cur, lin.weight.data, lin.bias.data, y, ops.ACT_NONE if last else self.act)
            if not last:
                self.h.append(y)
            cur = y
        return cur

    def backward(self, dy, ws, dx_out=None, x_is_activation=False):
        """dy: d loss / d output.  Fills self.grads; returns d loss / d input if dx_out is given
        (x_is_activation: the chain input is itself a tanh output whose derivative must be applied)."""
        n = len(self.linears)
        for i in reversed(range(n)):
            lin = self.linears[i]
            inp = self.h[i - 1] if i > 0 else self.x
            dW, db = self.grads[i]
            ops.linear_bwd_weight(dy, inp, dW, db, ws)
            if i > 0:
                dx = torch.empty_like(inp)
                ops.linear_bwd_data(dy, lin.weight.data, inp, dx, self.act)
                dy = dx
            elif dx_out is not None:
                ops.linear_bwd_data(dy, lin.weight.data, inp if x_is_activation else None, dx_out,
                                    self.act if x_is_activation else ops.ACT_NONE)
        return dx_out


class _HipNet(nn.Module):
    """Common plumbing: grad views + scratch workspace."""

    def _workspace(self, device):
        ws = getattr(self, "_ws", None)
        if ws is None or ws.device != device:
            ws = ops.Workspace(device)
            object.__setattr__(self, "_ws", ws)
        return ws

    def set_grad_views(self, views):
        """views: dict param-name -> tensor view into the owner's flat gradient buffer."""
        raise NotImplementedError

    def forward(self, x):                 # rollout / eval inference (no gradient state kept)
        with torch.no_grad():
            return self.hip_forward(x)


class MLP(_HipNet):
    """network.py:27-54: Linear(O,h0)-act-...-Linear(h_last,out), orthogonal init with gains
    sqrt(2)...,(1 for a scalar head | 0.01 for a policy head)."""

    def __init__(self, input_dim, output_dim, net_cfg, proprio_shape):
        super().__init__()
        hidden_dim = net_cfg['hid_dim']
        layers = [nn.Linear(input_dim, hidden_dim[0]), get_activation(net_cfg['activation'])]
        for l in range(len(hidden_dim)):
            if l == len(hidden_dim) - 1:
                layers.append(nn.Linear(hidden_dim[l], output_dim))
            else:
                layers.append(nn.Linear(hidden_dim[l], hidden_dim[l + 1]))
                layers.append(get_activation(net_cfg['activation']))
        self.model = nn.Sequential(*layers)
        self.output_dim = output_dim
        gains = [math.sqrt(2)] * len(hidden_dim) + [1 if output_dim == 1 else 0.01]
        lins = [m for m in self.model if isinstance(m, nn.Linear)]
        for g, m in zip(gains, lins):
            torch.nn.init.orthogonal_(m.weight, gain=g)
        object.__setattr__(self, "_chain", _LinearChain(lins, _act_code(net_cfg['activation'])))

    def set_grad_views(self, views):
        idx = [i for i, m in enumerate(self.model) if isinstance(m, nn.Linear)]
        self._chain.grads = [(views[f"model.{i}.weight"], views[f"model.{i}.bias"]) for i in idx]

    def hip_forward(self, x, out=None):
        return self._chain.forward(x, out)

    def hip_backward(self, dy):
        self._chain.backward(dy, self._workspace(dy.device))


class PointNet(_HipNet):
    """network.py:141-198: shared per-point MLP C->128->256->512 (act,act,none) -> max [| mean]
    pooling over the 1024 points -> (+proprio) -> 128 -> 32 -> out.  The per-point MLP and
    the pooling are ONE HIP kernel (pm_pointnet_enc_fwd_f32); the (B,1024,512) activation never
    exists.  `point_num` stays 1024 as in the reference (network.py:146) unless
    net_cfg['point_num'] overrides it (multiple of 64, <= 1024)."""

    def __init__(self, input_dim, output_dim, net_cfg, proprio_shape):
        super().__init__()
        self.max_mean_concat = net_cfg['max_mean']
        self.point_num = int(net_cfg.get('point_num', 1024))
        act = net_cfg['activation']
        self.mlp = nn.Sequential(
            nn.Linear(input_dim // self.point_num, 128), get_activation(act),
            nn.Linear(128, 256), get_activation(act),
            nn.Linear(256, 512),
        )
        self.final_mlp = nn.Sequential(
            nn.Linear(512 * (1 + self.max_mean_concat) + proprio_shape, 128), get_activation(act),
            nn.Linear(128, 32), get_activation(act),
            nn.Linear(32, output_dim),
        )
        self.proprio_shape = proprio_shape
        self.substract_mean = net_cfg['sub_mean']
        self.count = 0
        self.in_channels = input_dim // self.point_num
        self.feat_dim = 512 * (1 + int(self.max_mean_concat))
        _act_code(act)
        object.__setattr__(self, "_head", _LinearChain([self.final_mlp[0], self.final_mlp[2], self.final_mlp[4]],
                                                       _act_code(act)))
        object.__setattr__(self, "_enc_grads", None)
        object.__setattr__(self, "_packed", None)

    def set_grad_views(self, views):
        self._head.grads = [(views[f"final_mlp.{i}.weight"], views[f"final_mlp.{i}.bias"]) for i in (0, 2, 4)]
        object.__setattr__(self, "_enc_grads", [views[f"mlp.{i}.{k}"] for i in (0, 2, 4) for k in ("weight", "bias")])

    def _pack(self, device):
        if self._packed is None or self._packed.device != device:
            object.__setattr__(self, "_packed", torch.empty(ops.pointnet_packed_elems(), device=device))
        ops.pointnet_pack(self.mlp[2].weight.data, self.mlp[4].weight.data, self._packed)
        return self._packed

    def hip_forward(self, x, out=None):
        B = x.shape[0]
        packed = self._pack(x.device)
        feat = torch.empty(B, self.feat_dim + self.proprio_shape, device=x.device)
        argmax = torch.empty(B, 512, dtype=torch.int32, device=x.device)
        ops.pointnet_enc_fwd(x, self.point_num, self.in_channels, self.substract_mean, self.mlp[0].weight.data,
                             self.mlp[0].bias.data, self.mlp[2].bias.data, self.mlp[4].bias.data, packed,
                             self.max_mean_concat, feat, argmax)
        if self.proprio_shape != 0:
            feat[:, self.feat_dim:].copy_(x[:, -self.proprio_shape:])     # network.py:166-168,193-194
        object.__setattr__(self, "_saved", (x, feat, argmax))
        return self._head.forward(feat, out)

    def hip_backward(self, dy):
        x, feat, argmax = self._saved
        ws = self._workspace(dy.device)
        dfeat = torch.empty_like(feat)
        self._head.backward(dy, ws, dx_out=dfeat)
        g = self._enc_grads
        ops.pointnet_enc_bwd(x, self.point_num, self.in_channels, self.substract_mean, self.mlp[0].weight.data,
                             self.mlp[0].bias.data, self.mlp[2].bias.data, self.mlp[4].weight.data, self._packed,
                             self.max_mean_concat, dfeat, argmax, g[0], g[1], g[2], g[3], g[4], g[5], ws)


def _out_of_scope(name):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(
                f"backbone '{name}' (reference network.py) is an image/TSDF student outside this build's hot-path "
                "scope (SURVEY.md §8f rank 4); MLP and PointNet are implemented")
    _Stub.__name__ = name
    return _Stub


Conv3DNet = _out_of_scope("Conv3DNet")
PoolConv3DNet = _out_of_scope("PoolConv3DNet")
ResNet = _out_of_scope("ResNet")
depthResNet = _out_of_scope("depthResNet")
