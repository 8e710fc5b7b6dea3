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

# network.py:7-24: every activation the reference's `get_activation` knows, as epilogues of the Linear kernels and in the fused
# PointNet encoder (tanh: the tuned kernels; the others: their generic instantiation) and in Conv3DNet (whose input-layer
# stencil is a tanh kernel: other activations run that layer as patch matrix + GEMM).  The plug-in backbones that are absent
# from the reference (PointNet2, SparseUNet) are tanh kernels -- every shipped cfg's activation.
_LINEAR_ACT = {"tanh": ops.ACT_TANH, "relu": ops.ACT_RELU, "crelu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "elu": ops.ACT_ELU,
               "selu": ops.ACT_SELU, "sigmoid": ops.ACT_SIGMOID}
_SUPPORTED_ACT = {"tanh": ops.ACT_TANH}
# Consecutive hidden layers of an MLP as ONE launch (csrc/gemm2_f32.hip: gemm2_chain_kernel, stripe-local hand-offs instead of
# kernel boundaries).  OPT-IN (PARTMANIP_CHAIN=1): bit-identical to the layer-by-layer launches and level with them on an idle
# chip (37.1 vs 37.1 us for the three forward layers of cfg 2, 29.7 vs 30.3 us for the two data gradients), but with the actor's
# and the critic's chains sharing the chip it LOSES: 1.98 M against 2.29 M env-steps/s at cfg 2 (round 3, alternating runs) --
# resident work-groups waiting at a stripe counter hold CU slots the other network's short kernels would have filled.
import os as _os
CHAIN_LAUNCH = _os.environ.get("PARTMANIP_CHAIN", "0") == "1"


def get_activation(act_name):
    """network.py:7-24 (module objects only keep the Sequential indices / state_dict keys aligned)."""
    table = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "lrelu": nn.LeakyReLU,
             "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}
    if act_name not in table:
        print("invalid activation function!")
        return None
    return table[act_name]()


def _act_code(name, linear_only=False):
    """PM_ACT_* code of a cfg's `activation`.  linear_only: the backbone is a chain of Linear kernels (MLP), which take
    the whole set; the fused encoders implement tanh."""
    table = _LINEAR_ACT if linear_only else _SUPPORTED_ACT
    if name not in table:
        raise NotImplementedError(f"activation '{name}': the fused encoder kernels of this backbone implement tanh (every "
                                  f"shipped cfg); the MLP backbone takes {sorted(_LINEAR_ACT)}; there is no fallback")
    return table[name]


class _LinearChain:
    """Forward/backward of Linear-act-...-Linear over explicit buffers (K4/K5 kernels)."""

    def __init__(self, linears, act_code, final_act=False):
        self.linears = linears            # list[nn.Linear]
        self.act = act_code
        self.final_act = final_act        # activation after the last Linear too (PointNet++ shared MLPs); the
                                          # caller then folds act' of the chain OUTPUT into the dy it passes back
        self.h = []                        # saved activation outputs of the hidden layers
        self.x_w = None                    # the chain input again, in rows padded for 16-byte loads (ops.padded_cols): weight gradient only
        self.grads = None                  # list[(dW view, db view)] set by ActorCritic.flatten()
        self._cws = None                   # stripe counters of the chained launches (one set per network: they run on two streams)
        self.col_blocks = None             # [(weight column start, end, input column start)]: the caller's rows hold the first layer's
                                           # input columns in another order (PointNet2 group-all: [features | xyz | 0] instead of [xyz | features])

    def _chain_ws(self, device):
        if self._cws is None or self._cws.device != device:
            self._cws = ops.Workspace(device)
        return self._cws

    def forward(self, x, out=None, x_w=None):
        n = len(self.linears)
        self.x, self.x_w = x, x_w
        self.h = []
        cur = x
        # rows wider than the first layer's fan-in (the caller zero-filled the extra columns so that K is a multiple of the
        # GEMM's 32-wide K-step: the LDS-DMA kernels instead of the register-staged one): a zero-padded copy of the weights
        self.w0p = None
        if x.shape[1] != self.linears[0].in_features or self.col_blocks is not None:
            w0 = self.linears[0].weight.data
            ext = getattr(self, "w0p_ext", None)           # kept current by the owner's operand-copy gather (PointNet2._arena_refresh)
            if ext is not None and ext.shape == (w0.shape[0], x.shape[1]) and ext.device == x.device:
                self.w0p = ext
            else:
                self.w0p = torch.empty(w0.shape[0], x.shape[1], device=x.device)
                ops.col_blocks(self.w0p, w0, self.col_blocks or [(0, w0.shape[1], 0)])    # one launch: permuted blocks + zero padding
        for i, lin in enumerate(self.linears):
            last = i == n - 1
            y = out if (last and out is not None) else torch.empty(cur.shape[0], lin.out_features, device=cur.device)
            ops.linear_fwd(cur, self.w0p if (i == 0 and self.w0p is not None) else lin.weight.data, lin.bias.data, y,
                           self.act if (not last or self.final_act) else ops.ACT_NONE)
            if not last:
                self.h.append(y)
            cur = y
        return cur

    def forward_hidden(self, x, x_w=None):
        """All layers but the last (the head runs fused with its loss: ops.ppo_actor_head); returns the last hidden activation.
        Small-step regime: the hidden layers run as ONE launch when their shapes allow (ops.linear_fwd_chain: the stripes of a
        layer hand their tiles to the next layer inside the launch), else layer by layer."""
        self.x, self.x_w = x, x_w
        self.h = []
        hidden = self.linears[:-1]
        if CHAIN_LAUNCH and len(hidden) >= 2 and x.shape[0] >= 256:
            ys = [torch.empty(x.shape[0], lin.out_features, device=x.device) for lin in hidden]
            # (the input layer reads the observations in their 16-byte-padded rows when the caller has them: same values)
            ins = [x_w if (x_w is not None and x_w.shape == x.shape) else x] + ys[:-1]
            items = [(i_, lin.weight.data, lin.bias.data, y, self.act) for i_, lin, y in zip(ins, hidden, ys)]
            if ops.linear_fwd_chain(items, self._chain_ws(x.device)):
                self.h = ys
                return ys[-1]
        cur = x
        for lin in hidden:
            y = torch.empty(cur.shape[0], lin.out_features, device=cur.device)
            ops.linear_fwd(cur, lin.weight.data, lin.bias.data, y, self.act)
            self.h.append(y)
            cur = y
        return cur

    def backward(self, dy, ws, dx_out=None, x_is_activation=False, dx_cols=None):
        """dy: d loss / d output.  Fills self.grads; returns d loss / d input if dx_out is given
        (x_is_activation: the chain input is itself a tanh output whose derivative must be applied; dx_cols: only the
        first dx_cols input columns are wanted -- dx_out is that narrow)."""
        n = len(self.linears)
        for i in reversed(range(n)):
            lin = self.linears[i]
            inp = self.h[i - 1] if i > 0 else self.x
            dW, db = self.grads[i]
            w0p = getattr(self, "w0p", None) if i == 0 else None
            if w0p is not None:                            # zero-padded K: the gradient of the real columns is the leading block
                dWp = torch.empty_like(w0p)
                ops.linear_bwd_weight(dy, inp, dWp, db, ws)
                # back in the parameter's column order, one launch (columns outside the blocks are padding: zero gradient)
                ops.col_blocks(dW, dWp, [(d_, d_ + e_ - s_, s_) for s_, e_, d_ in (self.col_blocks or [(0, dW.shape[1], 0)])])
            else:
                ops.linear_bwd_weight(dy, inp, dW, db, ws)
            if i > 0:
                dx = torch.empty_like(inp)
                ops.linear_bwd_data(dy, lin.weight.data, inp, dx, self.act)
                dy = dx
            elif dx_out is not None:
                w_in = w0p if w0p is not None else lin.weight.data
                ops.linear_bwd_data(dy, w_in if dx_cols is None else w_in[:, :dx_cols], inp if x_is_activation else None, dx_out,
                                    self.act if x_is_activation else ops.ACT_NONE)
        return dx_out


def chains_forward(chains, xs):
    """Forward of several _LinearChains with the same depth, ONE grouped launch per layer (ops.linear_fwd_group): the actor's
    and the critic's layer l are independent problems (ppo.py:73-74) and share a grid.  Same arithmetic as
    `_LinearChain.forward` per chain."""
    n = len(chains[0].linears)
    assert all(len(ch.linears) == n for ch in chains)
    cur = list(xs)
    for ch, x in zip(chains, xs):
        ch.x, ch.x_w, ch.h = x, None, []
    for i in range(n):
        last = i == n - 1
        items = []
        for ch, c in zip(chains, cur):
            lin = ch.linears[i]
            y = torch.empty(c.shape[0], lin.out_features, device=c.device)
            items.append((c, lin.weight.data, lin.bias.data, y, ch.act if (not last or ch.final_act) else ops.ACT_NONE))
        ops.linear_fwd_group(items)
        cur = [it[3] for it in items]
        if not last:
            for ch, y in zip(chains, cur):
                ch.h.append(y)
    return cur


def chains_backward(chains, dys, slab_strides, splits, head_dz=None):
    """Backward of the same chains: one grouped data-gradient launch per layer, then ALL weight gradients of all chains in
    (at most) two grouped launches.  splits > 1: the reduction over the batch is cut into `splits` slabs written at
    grad + z * slab_stride (slab 0 = the chain's gradient views); the optimiser launch sums them (ops.clip_adam_group).
    head_dz: the heads' data gradients (d loss / d pre-activation of the last hidden layer) when the caller's fused head
    launch has already produced them."""
    n = len(chains[0].linears)
    dy = list(dys)
    dz = [[None] * n for _ in chains]                      # dz[c][i] = d loss / d (pre-activation of layer i)
    for c, d in enumerate(dy):
        dz[c][n - 1] = d
    top = n
    if head_dz is not None:
        for c, d in enumerate(head_dz):
            dz[c][n - 2] = d
        top = n - 1
    if CHAIN_LAUNCH and len(chains) == 1 and top - 1 >= 2 and dz[0][top - 1].shape[0] >= 256 and dz[0][top - 1].shape[1] > 16:
        # one network's hidden-layer data gradients as ONE launch (ops.linear_bwd_data_chain), top layer first
        ch = chains[0]
        items, d = [], dz[0][top - 1]
        for i in reversed(range(1, top)):
            inp = ch.h[i - 1]
            dx = torch.empty_like(inp)
            items.append((d, ch.linears[i].weight.data, inp, dx, ch.act))
            d = dx
        if ops.linear_bwd_data_chain(items, ch._chain_ws(d.device)):
            for i, it in zip(reversed(range(1, top)), items):
                dz[0][i - 1] = it[3]
            top = 1
    for i in reversed(range(1, top)):
        items = []
        for c, ch in enumerate(chains):
            inp = ch.h[i - 1]
            items.append((dz[c][i], ch.linears[i].weight.data, inp, torch.empty_like(inp), ch.act))
        if len(items) == 1 and items[0][0].shape[1] <= 16:
            # a lone policy / value head: the single-problem entry point takes its row-wise VALU kernel (5.8 us against 12.3
            # as a 64-wide GEMM tile with 10 live columns)
            it = items[0]
            ops.linear_bwd_data(it[0], it[1], it[2], it[3], it[4])
        else:
            ops.linear_bwd_data_group(items)
        for c, it in enumerate(items):
            dz[c][i - 1] = it[3]
    vec, rest = [], []
    for c, ch in enumerate(chains):
        for i in range(n):
            inp = ch.h[i - 1] if i > 0 else (ch.x_w if ch.x_w is not None else ch.x)
            dW, db = ch.grads[i]
            item = (dz[c][i], inp, dW, db, slab_strides[c])
            # 16-byte loads: whole float4s per row, counting the readable padding of ops.padded_cols buffers
            ncol, kcol = getattr(dz[c][i], "_pm_cols", dz[c][i].shape[1]), getattr(inp, "_pm_cols", inp.shape[1])
            ok = ncol % 4 == 0 and kcol % 4 == 0 and inp.stride(0) % 4 == 0 and inp.data_ptr() % 16 == 0 and \
                dz[c][i].stride(0) % 4 == 0 and dz[c][i].data_ptr() % 16 == 0
            (vec if ok else rest).append(item)
    for grp in (vec, rest):                                # 16-byte-loadable problems must not share a launch with the others
        for lo in range(0, len(grp), 8):
            ops.linear_bwd_weight_group(grp[lo:lo + 8], splits)


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
        object.__setattr__(self, "_chain", _LinearChain(lins, _act_code(net_cfg['activation'], linear_only=True)))

    def set_grad_views(self, views):
        idx = [i for i, m in enumerate(self.model) if isinstance(m, nn.Linear)]
        self._chain.grads = [(views[f"model.{i}.weight"], views[f"model.{i}.bias"]) for i in idx]

    def hip_forward(self, x, out=None, x_w=None):
        return self._chain.forward(x, out, x_w)

    def hip_backward(self, dy):
        self._chain.backward(dy, self._workspace(dy.device))


class PointNet(_HipNet):
    """network.py:141-198: shared per-point MLP C->128->256->512 (act,act,none) -> max [| mean]
    pooling over the 1024 points -> (+proprio) -> 128 -> 32 -> out.  The per-point MLP and
    the pooling are ONE HIP kernel (pm_pointnet_enc_fwd_f32); the (B,1024,512) activation never
    exists.  `point_num` stays 1024 as in the reference (network.py:146) unless
    net_cfg['point_num'] overrides it (multiple of 64, <= 4096: BASELINE cfg 5's 4096-point clouds)."""

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
        # 'f32' (default): exact fp32 MFMA.  Opt-in (not reference keys), encoder forward on split-bf16 MFMAs, the
        # backward stays fp32: 'bf16x3' (two planes, ~1e-5 relative, csrc/pointnet_enc_bf3.hip) and 'bf16x6' (three
        # planes, six products: the fp32 kernel's error level, csrc/pointnet_enc_bf6.hip).
        self.precision = net_cfg.get('precision', 'f32')
        self.save_h2 = bool(net_cfg.get('save_h2', True))
        if self.precision not in ('f32', 'bf16x3', 'bf16x6'):
            raise ValueError(f"PointNet precision '{self.precision}'")
        # 'precision_bwd': 'f32' (default) | 'bf16x6' -- the backward's two dense GEMMs on three-plane split-bf16 MFMAs
        # (csrc/pointnet_enc_bwd_bf6.h: the fp32 kernel's error level); needs the saved layer 2 and tanh
        self.precision_bwd = net_cfg.get('precision_bwd', 'f32')
        if self.precision_bwd not in ('f32', 'bf16x6'):
            raise ValueError(f"PointNet precision_bwd '{self.precision_bwd}'")
        if self.precision_bwd != 'f32' and not self.save_h2:
            raise ValueError("PointNet precision_bwd 'bf16x6' reads the forward's saved layer 2 (save_h2: True)")
        # every activation of get_activation (network.py:7-24): tanh runs the tuned packed-tanh encoder kernels, the others
        # their generic instantiation (pm_act / pm_dact); the split-bf16 forwards are tanh kernels
        code = _act_code(act, linear_only=True)
        if (self.precision != 'f32' or self.precision_bwd != 'f32') and code != ops.ACT_TANH:
            raise NotImplementedError(f"PointNet precision '{self.precision}' / precision_bwd '{self.precision_bwd}' is a tanh "
                                      f"kernel; activation '{act}' runs on 'f32'")
        object.__setattr__(self, "_act", code)
        object.__setattr__(self, "_head", _LinearChain([self.final_mlp[0], self.final_mlp[2], self.final_mlp[4]], code))
        object.__setattr__(self, "_enc_grads", None)
        object.__setattr__(self, "_packed", None)
        object.__setattr__(self, "_packed3", None)
        object.__setattr__(self, "_packed6", None)
        object.__setattr__(self, "_packed6b", None)

    def set_grad_views(self, views):
        self._head.grads = [(views[f"final_mlp.{i}.weight"], views[f"final_mlp.{i}.bias"]) for i in (0, 2, 4)]
        object.__setattr__(self, "_enc_grads", [views[f"mlp.{i}.{k}"] for i in (0, 2, 4) for k in ("weight", "bias")])

    def _pack(self, device):
        if self._packed is None or self._packed.device != device:
            object.__setattr__(self, "_packed", torch.empty(ops.pointnet_packed_elems(), device=device))
        ops.pointnet_pack(self.mlp[2].weight.data, self.mlp[4].weight.data, self._packed)
        return self._packed

    def forward(self, x):                 # rollout / eval inference: nothing is kept for a backward
        with torch.no_grad():
            return self.hip_forward(x, save_h2=False)

    def _h2_buffer(self, B, device):
        """(B, P, 256) fp32, reused across steps: the training forward stores the layer-2 activations here and the
        backward loads them instead of recomputing layer 2 (2 GB at 2048 clouds x 1024 points; HBM is 288 GB and
        the kernels are MFMA-bound, so the extra 1 KB per point each way is free and a third of the backward's
        matrix work disappears).  `net_cfg['save_h2']: False` restores the recompute path."""
        n = B * self.point_num * 256
        buf = getattr(self, "_h2buf", None)
        if buf is None or buf.numel() < n or buf.device != device:
            buf = torch.empty(n, device=device)
            object.__setattr__(self, "_h2buf", buf)
        return buf[:n].view(B, self.point_num, 256)

    def hip_forward(self, x, out=None, save_h2=None):
        B = x.shape[0]
        packed = self._pack(x.device)
        feat = torch.empty(B, self.feat_dim + self.proprio_shape, device=x.device)
        argmax = torch.empty(B, 512, dtype=torch.int32, device=x.device)
        save_h2 = self.save_h2 if save_h2 is None else save_h2
        h2 = self._h2_buffer(B, x.device) if save_h2 else None
        if self.precision == 'bf16x3':
            if self._packed3 is None or self._packed3.device != x.device:
                object.__setattr__(self, "_packed3", torch.empty(int(ops.lib.pm_pointnet_packed_bf3_bytes()),
                                                                 dtype=torch.uint8, device=x.device))
            ops.pointnet_pack_bf3(self.mlp[2].weight.data, self.mlp[4].weight.data, self._packed3)
            ops.pointnet_enc_fwd_bf3(x, self.point_num, self.in_channels, self.substract_mean, self.mlp[0].weight.data,
                                     self.mlp[0].bias.data, self.mlp[2].bias.data, self.mlp[4].bias.data,
                                     self._packed3, self.max_mean_concat, feat, argmax, h2)
        elif self.precision == 'bf16x6':
            if self._packed6 is None or self._packed6.device != x.device:
                object.__setattr__(self, "_packed6", torch.empty(int(ops.lib.pm_pointnet_packed_bf6_bytes()),
                                                                 dtype=torch.uint8, device=x.device))
            ops.pointnet_pack_bf6(self.mlp[2].weight.data, self.mlp[4].weight.data, self._packed6)
            ops.pointnet_enc_fwd_bf6(x, self.point_num, self.in_channels, self.substract_mean, self.mlp[0].weight.data,
                                     self.mlp[0].bias.data, self.mlp[2].bias.data, self.mlp[4].bias.data,
                                     self._packed6, self.max_mean_concat, feat, argmax, h2)
        else:
            ops.pointnet_enc_fwd(x, self.point_num, self.in_channels, self.substract_mean, self.mlp[0].weight.data,
                                 self.mlp[0].bias.data, self.mlp[2].bias.data, self.mlp[4].bias.data, packed,
                                 self.max_mean_concat, feat, argmax, h2, self._act)
        if self.proprio_shape != 0:
            feat[:, self.feat_dim:].copy_(x[:, -self.proprio_shape:])     # network.py:166-168,193-194
        object.__setattr__(self, "_saved", (x, feat, argmax, h2))
        return self._head.forward(feat, out)

    def hip_backward(self, dy):
        x, feat, argmax, h2 = self._saved
        ws = self._workspace(dy.device)
        dfeat = torch.empty_like(feat)
        self._head.backward(dy, ws, dx_out=dfeat)
        g = self._enc_grads
        if self.precision_bwd == 'bf16x6':
            if h2 is None:
                raise RuntimeError("PointNet precision_bwd 'bf16x6': the forward did not save layer 2 (an inference forward?)")
            if self._packed6b is None or self._packed6b.device != dy.device:
                object.__setattr__(self, "_packed6b", torch.empty(int(ops.lib.pm_pointnet_packed_bwd_bf6_bytes()),
                                                                  dtype=torch.uint8, device=dy.device))
            ops.pointnet_pack_bwd_bf6(self.mlp[2].weight.data, self._packed6b)
            ops.pointnet_enc_bwd_bf6(x, self.point_num, self.in_channels, self.substract_mean, self.mlp[0].weight.data,
                                     self.mlp[0].bias.data, self.mlp[2].bias.data, self.mlp[4].weight.data, self._packed,
                                     self._packed6b, self.max_mean_concat, dfeat, argmax, g[0], g[1], g[2], g[3], g[4], g[5],
                                     ws, h2)
            return
        ops.pointnet_enc_bwd(x, self.point_num, self.in_channels, self.substract_mean, self.mlp[0].weight.data,
                             self.mlp[0].bias.data, self.mlp[2].bias.data, self.mlp[4].weight.data, self._packed,
                             self.max_mean_concat, dfeat, argmax, g[0], g[1], g[2], g[3], g[4], g[5], ws, h2, self._act)


def _pad4(n):
    return (n + 3) // 4 * 4


class _GeomTabs(list):
    """Per-level (centres, neighbour index) tables of a rollout + the packed-row plans of the mini-batch slices taken from it."""

    def __init__(self, it=()):
        super().__init__(it)
        self.plans = {}


class PointNet2(_HipNet):
    """PointNet++ (single-scale grouping) encoder as a backbone plug-in (`network.name: PointNet2`).

    NOT in the reference snapshot (README.md:23,30 -- the paper's point-cloud backbone moved to an
    unmounted branch); BASELINE.json's north_star mandates it, so this follows the published
    PointNet++ SSG structure with the reference's conventions (no BatchNorm, `net_cfg['activation']`,
    the PointNet head 128-32-out, proprio appended before the head):
        SA level l: farthest-point-sample `npoints[l]` centres (K12) -> ball query `radii[l]`,
        `nsamples[l]` (K13) -> rows [xyz - centre | features] (K15) -> shared per-point MLP
        `mlps[l]` (Linear kernels, activation after every layer) -> max over the group (K15);
        last entry of `mlps` = group-all level on absolute coordinates -> global max.
    Inputs to each shared MLP are zero-padded to a multiple of 4 channels (16-B rows), so the first
    Linear of a level has `pad4(3 + C_prev)` input features; the pad columns never receive data."""

    def __init__(self, input_dim, output_dim, net_cfg, proprio_shape):
        super().__init__()
        self.point_num = int(net_cfg.get('point_num', 1024))
        self.in_channels = input_dim // self.point_num
        self.proprio_shape = proprio_shape
        self.npoints = list(net_cfg.get('npoints', [256, 64]))
        self.radii = list(net_cfg.get('radii', [0.2, 0.4]))
        self.nsamples = list(net_cfg.get('nsamples', [32, 32]))
        mlps = [list(m) for m in net_cfg.get('mlps', [[64, 64, 128], [128, 128, 256], [256, 512]])]
        assert len(mlps) == len(self.npoints) + 1 == len(self.radii) + 1 == len(self.nsamples) + 1
        if not self.npoints:
            raise ValueError("PointNet2 needs at least one set-abstraction level (npoints / radii / nsamples / mlps[:-1]); a network "
                             "with only the group-all level is `PointNet`")
        counts = [self.point_num] + self.npoints
        if any(b > a or b < 1 for a, b in zip(counts, counts[1:])):
            raise ValueError(f"PointNet2: npoints {self.npoints} must be non-increasing and <= point_num {self.point_num} "
                             "(a level cannot sample more centres than it has points)")
        act = net_cfg['activation']
        code = _act_code(act)
        cf = self.in_channels - 3
        assert cf >= 0, "PointNet2 needs xyz as the first three channels"
        self.sa = nn.ModuleList()
        self.in_feats = []
        chains = []
        for dims in mlps:
            cin = _pad4(3 + cf)
            self.in_feats.append(cf)
            layers, lins = [], []
            for d in dims:
                lin = nn.Linear(cin, d)
                layers += [lin, get_activation(act)]
                lins.append(lin)
                cin = d
            self.sa.append(nn.Sequential(*layers))
            chains.append(_LinearChain(lins, code, final_act=True))
            cf = dims[-1]
        self.feat_dim = cf
        self.final_mlp = nn.Sequential(
            nn.Linear(cf + proprio_shape, 128), get_activation(act),
            nn.Linear(128, 32), get_activation(act),
            nn.Linear(32, output_dim),
        )
        object.__setattr__(self, "_chains", chains)
        object.__setattr__(self, "_head", _LinearChain([self.final_mlp[0], self.final_mlp[2], self.final_mlp[4]], code))
        # levels that run as ONE fused kernel per direction (pm_sa_fwd_f32 / pm_sa_bwd_f32); the others use the
        # separate gather / Linear / max-pool kernels (still HIP, just unfused)
        fused_ok = bool(net_cfg.get('fused_sa', True))
        object.__setattr__(self, "_fused", [
            fused_ok and len(mlps[l]) == 3 and ops.sa_supported(*mlps[l], self.nsamples[l])
            for l in range(len(self.npoints))])
        # group-all level: its last layer runs fused with the max over the cloud, the backward without the two dense GEMMs on
        # the one-non-zero-per-(cloud, channel) gradient (csrc/sa_groupall.hip); `fused_groupall: False` keeps Linear + max-pool (A/B)
        ga = mlps[-1]
        object.__setattr__(self, "_ga_fused", bool(
            fused_ok and net_cfg.get('fused_groupall', True) and len(ga) >= 2 and act == 'tanh'
            and bool(self.npoints) and ops.sa_groupall_supported(ga[-2], ga[-1], self.npoints[-1])))
        object.__setattr__(self, "_ga_chain", _LinearChain(chains[-1].linears[:-1], code, final_act=True) if self._ga_fused else None)
        object.__setattr__(self, "_ga_packed", None)
        # ... and the last set-abstraction level then writes its pooled rows straight into the group-all input rows
        # ([features | xyz | 0], the first layer's weight columns permuted alike): no gather copy forward, no scatter copy backward
        object.__setattr__(self, "_ga_direct", bool(self._ga_fused and self.npoints and self._fused[-1]
                                                    and net_cfg.get('groupall_direct_rows', True)))
        if self._ga_direct:
            self._ga_chain.col_blocks = [(3, 3 + mlps[-2][-1], 0), (0, 3, mlps[-2][-1])]
        object.__setattr__(self, "_sa_packed", [None] * len(self.npoints))
        object.__setattr__(self, "_sa_h2", [None] * len(self.npoints))
        object.__setattr__(self, "_save_h2_now", False)
        self.save_h2 = bool(net_cfg.get('save_h2', True))
        # fused levels run over each group's DISTINCT rows (ball query pads short groups with copies of their first hit;
        # a copy never wins the max-pool): same outputs and gradients as the dense kernels, `False` keeps the dense form (A/B)
        self.unique_rows = bool(net_cfg.get('sa_unique_rows', True))
        # the gradient of a level's per-source-point layer-1 rows summed in a FIXED order over the plan's inverse table (no fp32
        # atomics: every bit of a backward is reproducible run to run); `False` keeps the atomic scatter (A/B)
        self.sa_deterministic = bool(net_cfg.get('sa_deterministic', True))
        object.__setattr__(self, "_sa_dz1", [None] * len(self.npoints))
        self.sa_fused_dy = bool(net_cfg.get('sa_fused_dy', True))             # False: segmented-sum pass + the two Linear launches (A/B)
        self.weight_arena = bool(net_cfg.get('weight_arena', True))           # False: one pack / copy launch per operand copy (A/B)
        object.__setattr__(self, "_arena", None)
        object.__setattr__(self, "_arena_views", None)
        object.__setattr__(self, "_sa_packed_w1f", [None] * len(self.npoints))
        object.__setattr__(self, "_sa_grads", None)

    def set_grad_views(self, views):
        for l, ch in enumerate(self._chains):
            ch.grads = [(views[f"sa.{l}.{2 * i}.weight"], views[f"sa.{l}.{2 * i}.bias"]) for i in range(len(ch.linears))]
        self._head.grads = [(views[f"final_mlp.{i}.weight"], views[f"final_mlp.{i}.bias"]) for i in (0, 2, 4)]
        if self._ga_chain is not None:
            self._ga_chain.grads = self._chains[-1].grads[:-1]

    # ---- operand copies of the weights: ONE gather per forward ------------------------------------------------------
    # Before round 6 every forward re-packed its weights with one tiny launch per copy (two pm_sa_pack_weights_f32, the group-all
    # pack, the aligned W1 feature block, the K-step-padded first group-all layer, the consumer's operand copy): six launches that
    # each queue behind the other network's persistent kernels (6 % of the step's kernel time in rocprofv3).  All of them are
    # gathers of the flat parameter buffer with a layout-fixed table: the table is recorded once by running those same entry
    # points on index-valued weights, and a forward refreshes the whole arena with pm_gather_copy_f32.
    def _arena_segments(self):
        """[(key, numel)] of the weight-derived copies this configuration uses (16-byte aligned segments of one buffer)."""
        segs = []
        for l in range(len(self.npoints)):
            if not self._fused[l]:
                continue
            lin1, lin2, lin3 = self.sa[l][0], self.sa[l][2], self.sa[l][4]
            dims = (lin1.out_features, lin2.out_features, lin3.out_features)
            segs.append((("sa_packed", l), int(ops.lib.pm_sa_packed_elems(*dims))))
            cf = self.in_feats[l]
            if cf > 0:
                segs.append((("w1f", l), dims[0] * cf))
                if ops.sa_dy_consume_supported(dims[0], cf):
                    segs.append((("dyc", l), int(ops.lib.pm_sa_dy_consume_packed_elems(dims[0], cf))))
        if self._ga_fused:
            lin = self._chains[-1].linears[-1]
            segs.append((("ga_packed",), int(ops.lib.pm_sa_groupall_packed_elems(lin.in_features, lin.out_features))))
            w0 = self._ga_chain.linears[0].weight
            ldo = (self.sa[-1][0].in_features + 31) // 32 * 32
            if ldo != w0.shape[1] or self._ga_chain.col_blocks is not None:
                segs.append((("ga_w0p",), w0.shape[0] * ldo))
        return segs

    def _arena_fill_by_entry_points(self, views):
        """Every copy through its own entry point (the pre-round-6 path; also what records the gather table)."""
        for key, v in views.items():
            if key[0] == "sa_packed":
                ops.sa_pack(self.sa[key[1]][2].weight.data, self.sa[key[1]][4].weight.data, v)
            elif key[0] == "w1f":
                cf = self.in_feats[key[1]]
                ops.col_blocks(v.view(-1, cf), self.sa[key[1]][0].weight.data, [(3, 3 + cf, 0)], zero_other=False)
            elif key[0] == "dyc":
                ops.sa_dy_consume_pack(self.sa[key[1]][0].weight.data, self.in_feats[key[1]], v)
            elif key[0] == "ga_packed":
                ops.sa_groupall_pack(self._chains[-1].linears[-1].weight.data, v)
            elif key[0] == "ga_w0p":
                w0 = self._ga_chain.linears[0].weight.data
                ops.col_blocks(v.view(w0.shape[0], -1), w0, self._ga_chain.col_blocks or [(0, w0.shape[1], 0)])

    def _arena_refresh(self, device):
        """Bring the operand copies up to date with the parameters; returns the views dict, or None when the parameters do not live
        in one flat buffer (a network used outside ActorCritic.flat()): the callers then pack per copy as before."""
        pf = getattr(self, "_param_flat", None)
        first = next(self.parameters())
        if not self.weight_arena or pf is None or pf.device != device or first.data_ptr() != pf.data_ptr() or pf.numel() >= (1 << 24):
            return None
        ar = getattr(self, "_arena", None)
        if ar is None or ar["src_ptr"] != pf.data_ptr() or ar["buf"].device != device:
            segs, off, views = self._arena_segments(), 0, {}
            if not segs:
                return None
            offs = []
            for key, n in segs:
                offs.append((key, off, n))
                off += (n + 63) // 64 * 64
            buf = torch.zeros(off, device=device)
            views = {key: buf[o:o + n] for key, o, n in offs}
            # record the table: parameters := their flat index + 1 (exact in fp32 below 2^24), run the entry points, read the indices back
            saved = pf.clone()
            pf.copy_(torch.arange(1, pf.numel() + 1, device=device, dtype=torch.float32))
            self._arena_fill_by_entry_points(views)
            table = buf.round().to(torch.int32) - 1                      # untouched / padding slots hold 0 -> -1 -> written as 0
            pf.copy_(saved)
            ar = dict(buf=buf, table=table.contiguous(), views=views, src_ptr=pf.data_ptr())
            object.__setattr__(self, "_arena", ar)
        ops.gather_copy(ar["buf"], pf, ar["table"])
        return ar["views"]

    def _sa_forward_fused(self, l, xyz, feat, centers, idx_g, pooled, plan_slot=None, tail_xyz=None):
        """One fused SA level.  Layer 1's feature part is applied per SOURCE point (Y) by the Linear kernel."""
        B, Pl = xyz.shape[0], xyz.shape[1]
        lin1, lin2, lin3 = self.sa[l][0], self.sa[l][2], self.sa[l][4]
        dims = (lin1.out_features, lin2.out_features, lin3.out_features)
        cf = 0 if feat is None else feat.shape[2]
        Y, w1f = None, None
        av = self._arena_views                              # operand copies refreshed by ONE gather at the top of hip_forward (or None)
        if cf > 0:
            # the feature columns of W1 start 12 bytes into a row: an aligned copy (64 KB) takes the 16-byte LDS-DMA loaders
            if av is not None:
                w1f = av[("w1f", l)].view(dims[0], cf)
            else:
                w1f = ops.col_blocks(torch.empty(dims[0], cf, device=xyz.device), lin1.weight.data, [(3, 3 + cf, 0)], zero_other=False)
            Y = torch.empty(B * Pl, dims[0], device=xyz.device)
            ops.linear_fwd(feat.reshape(B * Pl, cf), w1f, None, Y, ops.ACT_NONE)
        if av is not None:
            packed = av[("sa_packed", l)]
        else:
            packed = self._sa_packed[l]
            if packed is None or packed.device != xyz.device:
                packed = torch.empty(int(ops.lib.pm_sa_packed_elems(*dims)), device=xyz.device)
                self._sa_packed[l] = packed
            ops.sa_pack(lin2.weight.data, lin3.weight.data, packed)
        h2 = None
        if self._save_h2_now:                              # training forward: keep layer 2 for the backward (reused buffer)
            n = idx_g.numel() * dims[1]
            buf = self._sa_h2[l]
            if buf is None or buf.numel() < n or buf.device != xyz.device:
                buf = self._sa_h2[l] = torch.empty(n, device=xyz.device)
            h2 = buf[:n]
        plan = None
        if self._uses_plan(l, centers.shape[1]):               # (more than 1024 groups per cloud: the padded kernels)
            cache, key = plan_slot if plan_slot is not None else (None, None)
            if key is not None:
                key = key + (dims,)                        # tile sizes follow the level's widths: actor and critic share a plan only when theirs agree
            plan = cache.get(key) if cache is not None else None
            if plan is None:
                plan = ops.sa_plan(idx_g, xyz, centers, dims, self._workspace(xyz.device), inverse=self._plan_inverse(l))
                if cache is not None:                      # a sequential mini-batch of a cached rollout: both networks and
                    cache[key] = plan.trim()               # every epoch reuse it (one host read of the row / tile counts)
                    plan.ready = torch.cuda.Event()
                    plan.ready.record()
            elif plan.ready is not None:                   # built on another stream (actor || critic): its tables must be complete
                torch.cuda.current_stream().wait_event(plan.ready)
            arg = ops.sa_fwd_packed(plan, Y, lin1.weight.data, lin1.bias.data, lin2.bias.data, lin3.bias.data,
                                    packed, dims, pooled, h2, tail_xyz=tail_xyz)
        else:
            arg = ops.sa_fwd(xyz, centers, idx_g, Y, lin1.weight.data, lin1.bias.data, lin2.bias.data, lin3.bias.data,
                             packed, dims, pooled, h2)
        return (idx_g, arg, "fused", xyz, feat, centers, Y, packed, dims, pooled, h2, plan, w1f, plan is not None and tail_xyz is not None)

    PLAN_BATCH = 4                        # mini-batch slices whose plans are built before one host read trims them

    def _uses_plan(self, l, S):
        """Does level l run over packed rows (pm_sa_plan_i32)?  (The plan's per-cloud pass holds up to 1024 groups.)"""
        return bool(self._fused[l] and self.unique_rows and S <= 1024)

    def _plan_inverse(self, l):
        """Does level l's plan carry the source point -> packed rows table (its layer-1 rows have a gradient to sum)?"""
        return bool(self.sa_deterministic and self.in_feats[l] > 0)

    def _sa_backward_fused(self, l, rec, dpooled, ws, need_dfeat):
        idx_g, arg, _, xyz, feat, centers, Y, packed, dims, pooled, h2, plan, w1f = rec[:13]
        B, Pl = xyz.shape[0], xyz.shape[1]
        lin1, lin2, lin3 = self.sa[l][0], self.sa[l][2], self.sa[l][4]
        (dW1, db1), (dW2, db2), (dW3, db3) = self._chains[l].grads
        cf = 0 if feat is None else feat.shape[2]
        det = cf > 0 and plan is not None and plan.inv_start is not None
        fused_dy = det and self.sa_fused_dy and ops.sa_dy_consume_supported(dims[0], cf)
        dY = None if (cf == 0 or fused_dy) else torch.empty(B * Pl, dims[0], device=xyz.device) if det else torch.zeros(B * Pl, dims[0], device=xyz.device)
        if det:
            n = plan.rowmap.shape[0] * dims[0]             # (R, C1) once the plan is trimmed, its capacity before
            buf = self._sa_dz1[l]
            if buf is None or buf.numel() < n or buf.device != xyz.device:
                buf = self._sa_dz1[l] = torch.empty(n, device=xyz.device)
            dz1 = buf[:n].view(-1, dims[0])
            ops.sa_bwd_packed(plan, Y, lin1.weight.data, lin1.bias.data, lin2.bias.data, lin3.weight.data,
                              packed, dims, pooled, arg, dpooled, dW1, db1, dW2, db2, dW3, db3, None, ws, h2, dz1=dz1)
            if fused_dy:
                # the sums are consumed where they are formed: dfeat = dY W1f and dW1[:, 3:3+cf] = dY^T feat in ONE launch, dY in LDS only
                av = self._arena_views
                if av is not None:
                    pw = av[("dyc", l)]
                else:
                    pw = self._sa_packed_w1f[l]
                    if pw is None or pw.device != xyz.device:
                        pw = self._sa_packed_w1f[l] = torch.empty(int(ops.lib.pm_sa_dy_consume_packed_elems(dims[0], cf)), device=xyz.device)
                    ops.sa_dy_consume_pack(lin1.weight.data, cf, pw)
                dfeat = torch.empty(B * Pl, cf, device=xyz.device) if need_dfeat else None
                return ops.sa_dy_consume(plan, dz1, feat.reshape(B * Pl, cf), pw, dfeat, dW1, ws)
            ops.sa_dy_segsum(plan, dz1, dY)
        elif plan is not None:
            ops.sa_bwd_packed(plan, Y, lin1.weight.data, lin1.bias.data, lin2.bias.data, lin3.weight.data,
                              packed, dims, pooled, arg, dpooled, dW1, db1, dW2, db2, dW3, db3, dY, ws, h2, zero_pad_cols=cf == 0)
        else:
            ops.sa_bwd(xyz, centers, idx_g, Y, lin1.weight.data, lin1.bias.data, lin2.bias.data, lin3.weight.data, packed,
                       dims, pooled, arg, dpooled, dW1, db1, dW2, db2, dW3, db3, dY, ws, h2)
        if cf == 0:
            if dW1.shape[1] > 3 and plan is None:          # (packed levels: the reduction launch zeroes them)
                ops.col_blocks(dW1, None, [], col0=3)      # pad columns never receive data (zero-only form)
            return None
        feat2 = feat.reshape(B * Pl, cf)
        dW1f = torch.empty_like(w1f)
        ops.linear_bwd_weight(dY, feat2, dW1f, None, ws)
        ops.col_blocks(dW1, dW1f, [(0, cf, 3)], col0=3)    # the feature columns' gradient in place, zeros in the pad columns
        if not need_dfeat:
            return None
        dfeat = torch.empty(B * Pl, cf, device=xyz.device)
        ops.linear_bwd_data(dY, w1f, None, dfeat, ops.ACT_NONE)
        return dfeat

    # ---- neighbourhood tables ------------------------------------------------------------------------
    # FPS centres and ball-query indices depend on the coordinates only -- not on any weight -- so a learner
    # that visits the same rollout rows in every epoch and for both networks computes them ONCE per rollout
    # (44 KB per cloud for the default levels: 1.4 GB for 4096 envs x 8 steps) and every forward takes rows of
    # these tables instead of re-running K12/K13 (11 % of the iteration before this).
    def precompute_geometry(self, obs, chunk=4096):
        """obs (M, O) -> [(centres (M,S_l,3) f32, idx (M,S_l,ns_l) i32) per SA level] for use_geometry()."""
        M, P, C = obs.shape[0], self.point_num, self.in_channels
        ws = self._workspace(obs.device)
        tabs = _GeomTabs((torch.empty(M, S, 3, device=obs.device),
                          torch.empty(M, S, self.nsamples[l], dtype=torch.int32, device=obs.device))
                         for l, S in enumerate(self.npoints))
        for lo in range(0, M, chunk):
            x = obs[lo:lo + chunk]
            b = x.shape[0]
            xyz = x[:, :P * C].reshape(b, P, C)[..., :3].contiguous()
            for l, S in enumerate(self.npoints):
                idx_c = ops.fps(xyz, S, ws)
                centers = ops.group_points(xyz, idx_c.view(b, S, 1)).view(b, S, 3)
                tabs[l][0][lo:lo + b] = centers
                tabs[l][1][lo:lo + b] = ops.ball_query(xyz, centers, self.radii[l], self.nsamples[l])
                xyz = centers
        return tabs

    def precompute_plans(self, tabs, obs, slices):
        """Packed-row plans (pm_sa_plan_i32) of every fused level for the (lo, n) row slices of `obs` a sequential pass will visit,
        built back to back and trimmed to their real sizes after ONE host read of all the row / tile counts (`_sa_forward_fused`
        builds a missing plan on first use with a read of its own).  Plans already in `tabs.plans` -- the other network's, same
        widths -- are kept."""
        P, C = self.point_num, self.in_channels
        ws = self._workspace(obs.device)
        built = []

        def flush():
            # plans are built at their worst-case capacity (24 B per padded row + the inverse table): trimming in bounded batches --
            # one host read per PLAN_BATCH slices -- keeps the transient at a few GB instead of all slices' capacities at once
            if not built:
                return
            totals = torch.stack([pl.totals for _, pl in built]).cpu()
            ready = torch.cuda.Event()
            for (key, pl), t in zip(built, totals):
                tabs.plans[key] = pl.trim((int(t[0]), int(t[1])))
                pl.ready = ready
            ready.record()
            built.clear()

        for k_, (lo, n) in enumerate(slices):
            xyz = None
            for l, S in enumerate(self.npoints):
                centers, idx_g = tabs[l][0][lo:lo + n], tabs[l][1][lo:lo + n]
                if self._uses_plan(l, S):
                    lin1, lin2, lin3 = self.sa[l][0], self.sa[l][2], self.sa[l][4]
                    dims = (lin1.out_features, lin2.out_features, lin3.out_features)
                    key = (l, lo, n, dims)
                    if key not in tabs.plans and key not in {k for k, _ in built}:
                        if xyz is None:
                            xyz = obs[lo:lo + n, :P * C].reshape(n, P, C)[..., :3].contiguous() if l == 0 else tabs[l - 1][0][lo:lo + n]
                        built.append((key, ops.sa_plan(idx_g, xyz, centers, dims, ws, inverse=self._plan_inverse(l))))
                xyz = centers
            if (k_ + 1) % self.PLAN_BATCH == 0:
                flush()
        flush()

    def use_geometry(self, tabs, rows):
        """Take the next forward's neighbourhood tables from `tabs`: rows = (lo, n) slice or an index tensor."""
        plans = getattr(tabs, "plans", None)
        if isinstance(rows, tuple):
            lo, n = rows
            sel = [(c[lo:lo + n], i[lo:lo + n], None if plans is None else (plans, (l, lo, n))) for l, (c, i) in enumerate(tabs)]
        else:
            r = rows.to(tabs[0][0].device, non_blocking=True)
            sel = [(c.index_select(0, r), i.index_select(0, r), None) for c, i in tabs]
        object.__setattr__(self, "_geom_next", sel)

    def forward(self, x):                 # rollout / eval inference: nothing is kept for a backward
        with torch.no_grad():
            return self.hip_forward(x, save_h2=False)

    def hip_forward(self, x, out=None, save_h2=None):
        B, P, C = x.shape[0], self.point_num, self.in_channels
        object.__setattr__(self, "_save_h2_now", self.save_h2 if save_h2 is None else save_h2)
        ws = self._workspace(x.device)
        pts = x[:, :P * C].reshape(B, P, C)
        xyz = pts[..., :3].contiguous()
        feat = pts[..., 3:].contiguous() if C > 3 else None
        geom = getattr(self, "_geom_next", None)
        object.__setattr__(self, "_geom_next", None)
        if geom is not None and geom[0][0].shape[0] != B:
            raise ValueError("use_geometry(): table rows do not match the batch")
        saved, ga_rows = [], None
        object.__setattr__(self, "_arena_views", self._arena_refresh(x.device))
        if self._arena_views is not None and ("ga_w0p",) in self._arena_views:
            w0 = self._ga_chain.linears[0].weight
            self._ga_chain.w0p_ext = self._arena_views[("ga_w0p",)].view(w0.shape[0], -1)
        elif self._ga_chain is not None:
            self._ga_chain.w0p_ext = None
        tail_done = False
        for l, S in enumerate(self.npoints):
            plan_slot = None
            if geom is not None:
                centers, idx_g, plan_slot = geom[l]
            else:
                idx_c = ops.fps(xyz, S, ws)
                centers = ops.group_points(xyz, idx_c.view(B, S, 1)).view(B, S, 3)
                idx_g = ops.ball_query(xyz, centers, self.radii[l], self.nsamples[l])
            if self._fused[l]:
                c3 = self.sa[l][4].out_features
                if self._ga_direct and l == len(self.npoints) - 1:
                    ga_rows = torch.empty(B * S, (self.sa[-1][0].in_features + 31) // 32 * 32, device=x.device)
                    pooled = ga_rows[:, :c3]
                else:
                    pooled = torch.empty(B * S, c3, device=x.device)
                direct_rows = self._ga_direct and l == len(self.npoints) - 1
                saved.append(self._sa_forward_fused(l, xyz, feat, centers, idx_g, pooled, plan_slot,
                                                    tail_xyz=centers.reshape(B * S, 3) if direct_rows else None))
                tail_done = tail_done or (direct_rows and saved[-1][13])
                xyz, feat = centers, pooled.view(B, S, -1)
                continue
            ldo = self.sa[l][0].in_features
            rows = ops.group_concat(xyz, feat, centers, idx_g, ldo)
            h = self._chains[l].forward(rows)
            pooled = torch.empty(B * S, h.shape[1], device=x.device)
            arg = ops.maxpool_rows(h, B * S, self.nsamples[l], pooled)
            saved.append((idx_g, arg, h, xyz.shape[1], 0 if feat is None else feat.shape[2], ldo))
            xyz, feat = centers, pooled.view(B, S, -1)
        S = xyz.shape[1]                                   # group-all level: absolute coordinates
        ldo = (self.sa[-1][0].in_features + 31) // 32 * 32     # zero columns up to the GEMM's K-step (the chain pads its weights alike)
        if ga_rows is not None:                            # the features are already in place: [features | xyz | 0]
            idx_all, rows, cf_ = None, ga_rows, feat.shape[2]
            if not tail_done:                              # (the packed level kernel writes [xyz | 0] behind its features itself)
                ops.col_blocks(rows, xyz.reshape(B * S, 3), [(0, 3, cf_)], col0=cf_)      # [features | xyz | 0]: the tail in one launch
        else:
            idx_all = torch.arange(S, dtype=torch.int32, device=x.device).repeat(B, 1).view(B, 1, S)
            zeros = torch.zeros(B, 1, 3, device=x.device)
            rows = ops.group_concat(xyz, feat, zeros, idx_all, ldo)
        fbuf = torch.empty(B, self.feat_dim + self.proprio_shape, device=x.device)
        if self._ga_fused:
            h = self._ga_chain.forward(rows)               # (B*S, CK): the layers before the last, tanh applied
            lin = self._chains[-1].linears[-1]
            if self._arena_views is not None:
                packed = self._arena_views[("ga_packed",)]
            else:
                packed = self._ga_packed
                if packed is None or packed.device != x.device:
                    packed = torch.empty(int(ops.lib.pm_sa_groupall_packed_elems(lin.in_features, lin.out_features)), device=x.device)
                    object.__setattr__(self, "_ga_packed", packed)
                ops.sa_groupall_pack(lin.weight.data, packed)
            arg = ops.sa_groupall_fwd(h, B, S, lin.bias.data, packed, fbuf[:, :self.feat_dim])
            saved.append((idx_all, arg, "groupall", S, feat.shape[2], ldo, h, fbuf))
        else:
            h = self._chains[-1].forward(rows)
            arg = ops.maxpool_rows(h, B, S, fbuf[:, :self.feat_dim])
            saved.append((idx_all, arg, h, S, feat.shape[2], ldo))
        if self.proprio_shape != 0:
            fbuf[:, self.feat_dim:].copy_(x[:, -self.proprio_shape:])
        object.__setattr__(self, "_saved", saved)
        return self._head.forward(fbuf, out)

    def hip_backward(self, dy):
        saved, B = self._saved, dy.shape[0]
        ws = self._workspace(dy.device)
        dfbuf = torch.empty(B, self.feat_dim + self.proprio_shape, device=dy.device)
        self._head.backward(dy, ws, dx_out=dfbuf)
        dpooled = dfbuf[:, :self.feat_dim]                 # (G, C) view with row stride feat_dim + proprio
        for l in reversed(range(len(saved))):
            if isinstance(saved[l][2], str) and saved[l][2] == "groupall":
                idx_g, arg, _, P_l, cf, ldo, h, fbuf = saved[l]
                lin = self._chains[-1].linears[-1]
                dW, db = self._chains[-1].grads[-1]
                dh = torch.empty_like(h)
                ops.sa_groupall_bwd(dpooled, fbuf[:, :self.feat_dim], arg, lin.weight.data, h, B, P_l, dh, dW, db, ws)
                direct = idx_g is None                     # the rows are [features | xyz | 0]: only the feature block's gradient is wanted
                drows = torch.empty(h.shape[0], cf if direct else ldo, device=dy.device) if l > 0 else None     # level-0 inputs are data
                self._ga_chain.backward(dh, ws, dx_out=drows, dx_cols=cf if (direct and l > 0) else None)
                if l > 0:
                    dpooled = drows if direct else ops.group_concat_bwd(drows, idx_g, B, P_l, cf, ldo).view(B * P_l, cf)
                continue
            if isinstance(saved[l][2], str):                 # fused level record
                # level-0 features are data: no gradient flows to them
                dpooled = self._sa_backward_fused(l, saved[l], dpooled, ws, need_dfeat=l > 0)
                continue
            idx_g, arg, h, P_l, cf, ldo = saved[l]
            ns = idx_g.shape[2]
            dh = ops.maxpool_rows_bwd(dpooled, arg, ns, y_tanh=h)      # max-pool + tanh' of the chain output
            need_dx = l > 0                                               # level-0 inputs are data, not activations
            drows = torch.empty(h.shape[0], ldo, device=dy.device) if need_dx else None
            self._chains[l].backward(dh, ws, dx_out=drows)
            if need_dx:
                dpooled = ops.group_concat_bwd(drows, idx_g, B, P_l, cf, ldo).view(B * P_l, cf)


class SparseUNet(_HipNet):
    """3D sparse-voxel U-Net encoder as a backbone plug-in (`network.name: SparseUNet`).

    The reference's README names a "3D Sparse-UNet" as the paper's vision backbone (README.md:30) but its code moved to an
    unmounted branch (README.md:23); BASELINE.json's north_star / config 5 ask for it, so this follows the published sparse
    U-Net structure (submanifold 3^3 convolutions, 2x strided levels, skip connections) with the reference's conventions
    (no BatchNorm, `net_cfg['activation']` = tanh, the PointNet head 128-32-out, proprio appended before the head) --
    PARITY UNPINNED; the test-side CPU restatement (`sparse_unet_forward`) is pinned to torch's dense conv3d U-Net on full grids.

    Input: the reference's 'depth_sparse' observation (tasks/hand_base.py:335-336, utils/depth2tsdf.py:88-120): `point_num`
    rows (x, y, z, f) per env with integer voxel coordinates in [0, grid), flattened (+ proprio tail).  Levels (channels
    c0, c1, c2; default 32, 64, 128; every Linear is followed by tanh):
        F0 = (f, x/grid, y/grid, z/grid)
        H0 = conv0(F0)  3^3 submanifold            D1 = down0(H0)  2^3 stride 2          H1 = conv1(D1)
        D2 = down1(H1)                              H2 = conv2(D2)
        E1 = up1([unpool(H2) | H1])                 E0 = up0([unpool(E1) | H0])            feat = max over the cloud's rows of E0
    A sparse convolution runs as a neighbour-row gather (pm_rows_gather_f32) + the fp32-MFMA Linear kernel on the
    (rows x J*C_in) matrix; geometry (dense index grids, neighbour / parent / child tables) is csrc/sparse_voxel.hip."""

    def __init__(self, input_dim, output_dim, net_cfg, proprio_shape):
        super().__init__()
        self.point_num = int(net_cfg.get('point_num', 1024))
        self.in_channels = input_dim // self.point_num
        if self.in_channels < 3:
            raise ValueError("SparseUNet needs rows (x, y, z[, f]) per point")
        self.grid = int(net_cfg.get('grid', 50))
        c0, c1, c2 = (int(c) for c in net_cfg.get('channels', [32, 64, 128]))
        if any(c % 4 for c in (c0, c1, c2)):
            raise ValueError("SparseUNet channels must be multiples of 4 (16-byte feature rows)")
        self.channels = (c0, c1, c2)
        self.proprio_shape = proprio_shape
        act = net_cfg['activation']
        code = _act_code(act)
        self.conv0, self.down0 = nn.Linear(27 * 4, c0), nn.Linear(8 * c0, c1)
        self.conv1, self.down1 = nn.Linear(27 * c1, c1), nn.Linear(8 * c1, c2)
        self.conv2 = nn.Linear(27 * c2, c2)
        self.up1, self.up0 = nn.Linear(c2 + c1, c1), nn.Linear(c1 + c0, c0)
        self.feat_dim = c0
        self.final_mlp = nn.Sequential(nn.Linear(c0 + proprio_shape, 128), get_activation(act), nn.Linear(128, 32),
                                       get_activation(act), nn.Linear(32, output_dim))
        object.__setattr__(self, "_act", code)
        object.__setattr__(self, "_head", _LinearChain([self.final_mlp[0], self.final_mlp[2], self.final_mlp[4]], code))
        object.__setattr__(self, "_g", None)
        self.fused_gather = bool(net_cfg.get('fused_gather', True))          # False: materialise every gathered operand (A/B)
        self.sparse_top = bool(net_cfg.get('sparse_top', True))               # False: conv2's weight gradient over all level-2 rows (A/B)

    _LAYERS = ("conv0", "down0", "conv1", "down1", "conv2", "up1", "up0")

    def set_grad_views(self, views):
        self._head.grads = [(views[f"final_mlp.{i}.weight"], views[f"final_mlp.{i}.bias"]) for i in (0, 2, 4)]
        object.__setattr__(self, "_g", {n: (views[f"{n}.weight"], views[f"{n}.bias"]) for n in self._LAYERS})

    def geometry(self, x):
        """Index grids and tables of the three levels for the clouds in x (coordinates only: no parameters involved)."""
        B, P, C, R = x.shape[0], self.point_num, self.in_channels, self.grid
        grid0, coords0, feat0 = ops.voxel_grid0(x, P, C, R)
        l1 = ops.voxel_down(coords0, grid0, R, B)
        l2 = ops.voxel_down(l1["coords"], l1["grid"], l1["R"], B)
        # conv0's operand is 27 taps x 4 channels = 108 columns: the level-0 table is built 32 taps wide (five absent taps), so
        # that the layer is four whole K-steps of the gathered GEMM without a padded copy of the table per forward
        pad0 = 32 if (self.fused_gather and self.in_channels >= 3) else 27
        nbr0p = ops.voxel_nbr27(coords0, grid0, R, pad0)
        return dict(feat0=feat0, nbr0=nbr0p[:, :27], nbr0p=nbr0p, nbr1=ops.voxel_nbr27(l1["coords"], l1["grid"], l1["R"]),
                    nbr2=ops.voxel_nbr27(l2["coords"], l2["grid"], l2["R"]), l1=l1, l2=l2, rows=(B * P, l1["rows"], l2["rows"]))

    @staticmethod
    def _mirror(nbr):
        """Mirrored neighbour table for the data gradient: row r reads s as neighbour j  <=>  s reads r as neighbour 26 - j,
        so the rows whose output used s are nbr[s].flip.  A row that is nobody's neighbour (a duplicate coordinate: its centre
        tap points at the canonical row, not at itself) gets -1 everywhere and a gradient of exactly zero."""
        return ops.voxel_mirror27(nbr)

    def _conv_dgrad(self, name, dz, nbr, y_in, dx):
        """dx = d loss / d (pre-activation of the layer that produced y_in) of the 3^3 convolution `name`, dz its output
        gradient: fused -- a gathered GEMM through the mirrored table, the (rows x 27*C_in) column gradient never exists --
        or, with fused_gather off, the Linear data gradient + the mirrored column gather."""
        lin = getattr(self, name)
        cout, cin = lin.weight.shape[0], lin.weight.shape[1] // 27
        if self.fused_gather and (27 * cout) % 32 == 0:
            wt = lin.weight.data.view(cout, 27, cin).permute(2, 1, 0).reshape(cin, 27 * cout).contiguous()
            g = self._saved["g"]
            key = "mirror_" + name                              # (kept with the geometry: cached tables are mirrored once)
            if key not in g:
                g[key] = self._mirror(nbr)
            return ops.sparse_conv_bwd_data(dz, g[key], wt, y_in, dx, self._act, self._zero(dz.device))
        dcols = torch.empty(dz.shape[0], 27 * cin, device=dz.device)
        ops.linear_bwd_data(dz, lin.weight.data, None, dcols, ops.ACT_NONE)
        return ops.rows_gather_bwd(dcols, nbr, cin, dx, reverse=True, self_col=13, y_tanh=y_in)

    def _lin(self, name, x, y):
        lin = getattr(self, name)
        ops.linear_fwd(x, lin.weight.data, lin.bias.data, y, self._act)
        return y

    def _conv(self, name, src, idx, C, y, idx_pad=None):
        """Sparse convolution `name` of the rows of `src` through the neighbour table `idx` (rows, J): fused -- the gather runs
        inside the GEMM's LDS-DMA loader and the (rows x J*C) operand never reaches HBM -- when J*C is a multiple of the K-step
        (every layer but conv0, whose 108-wide operand is small); returns the materialised operand or None."""
        lin = getattr(self, name)
        if self.fused_gather and (idx.shape[1] * C) % 32 == 0:
            ops.sparse_conv_fwd(src, idx, C, lin.weight.data, lin.bias.data, y, self._act, self._zero(src.device))
            return None
        pad = (-idx.shape[1] * C) % 32
        if self.fused_gather and pad % C == 0:
            # conv0: 27 taps x 4 input channels = 108 columns; five absent taps (index -1 -> zero rows) and zero weight columns
            # make it 128 = four K-steps, and the layer runs fused like the others.  Returns the padded table for the
            # weight gradient (geometry() builds the level-0 table that wide: `idx_pad`).
            idx_p = idx_pad if (idx_pad is not None and idx_pad.shape[1] == idx.shape[1] + pad // C) else \
                torch.nn.functional.pad(idx, (0, pad // C), value=-1)
            w_p = torch.zeros(lin.weight.shape[0], idx_p.shape[1] * C, device=src.device)
            w_p[:, :lin.weight.shape[1]].copy_(lin.weight.data)
            ops.sparse_conv_fwd(src, idx_p, C, w_p, lin.bias.data, y, self._act, self._zero(src.device))
            return idx_p
        cols = ops.rows_gather(src, idx, C, torch.empty(idx.shape[0], idx.shape[1] * C, device=src.device))
        self._lin(name, cols, y)
        return cols

    def _conv_wgrad(self, name, dz, src, idx, C, cols, ws):
        dW, db = self._g[name]
        if cols is None:
            ops.sparse_conv_bwd_weight(dz, src, idx, C, dW, db, self._zero(src.device), ws)
        elif cols.dtype == torch.int32:                       # the padded table of a fused layer with a ragged K (conv0)
            dW_p = torch.empty(dW.shape[0], cols.shape[1] * C, device=src.device)
            ops.sparse_conv_bwd_weight(dz, src, cols, C, dW_p, db, self._zero(src.device), ws)
            dW.copy_(dW_p[:, :dW.shape[1]])
        else:
            ops.linear_bwd_weight(dz, cols, dW, db, ws)

    def _zero(self, device):
        z = getattr(self, "_zero_row", None)
        if z is None or z.device != device:
            z = torch.zeros(max(self.channels) + 64, device=device)
            object.__setattr__(self, "_zero_row", z)
        return z

    def forward(self, x):
        with torch.no_grad():
            return self.hip_forward(x)

    def hip_forward(self, x, out=None):
        with ops.TIMER.bracket("sparse_unet_fwd"):
            return self._hip_forward(x, out)

    def hip_backward(self, dy):
        with ops.TIMER.bracket("sparse_unet_bwd"):
            self._hip_backward(dy)

    def _vcat_ok(self):
        """The concatenated operands of up1 / up0 ([unpool(H2) | H1], [unpool(E1) | H0]) stay VIRTUAL when the coarse width is a
        multiple of the skip width: both halves then live in one buffer of skip-width rows (the coarse matrix viewed as m rows
        per voxel, the skip rows behind it) and the GEMM loader gathers (m + 1) 'taps' per output row -- no un-pooling copy, no
        concatenated matrix in HBM.  fused_gather off, or other widths: the materialised form."""
        c0, c1, c2 = self.channels
        return self.fused_gather and c2 % c1 == 0 and c1 % c0 == 0 and (c2 + c1) % 32 == 0 and (c1 + c0) % 32 == 0

    @staticmethod
    def _vcat_table(parent, m, rows_hi):
        """(rows, m + 1) gather table of a virtual [unpool | skip] operand: m chunks of the parent's row, then the row's own."""
        return ops.voxel_vcat_table(parent.view(-1), m, rows_hi)

    # ---- geometry once per rollout (ppo.update): the tables depend on the coordinates only, and a sequential mini-batch is the
    # same slice of the rollout in every epoch and for both networks -- 10 forwards share one set of tables (and the two host
    # reads that size the strided levels happen once per mini-batch instead of once per forward)
    def precompute_geometry(self, obs):
        return dict(cache={}, rows=obs.shape[0])

    def use_geometry(self, geom, rows):
        """Take the next forward's tables from `geom` (built on first use) when `rows` is a (lo, n) slice of the rollout."""
        object.__setattr__(self, "_geom_next", (geom["cache"], rows) if isinstance(rows, tuple) else None)

    def take_geometry(self, g):
        """The next forward's tables, built ahead by the caller from the SAME rows (dagger.update: geometry(x) of mini-batch
        k + 1 on a side stream under the GEMMs of mini-batch k -- the tables are latency-bound hash lookups, the GEMMs MFMA-bound)."""
        object.__setattr__(self, "_geom_ready", g)

    def _hip_forward(self, x, out=None):
        B, P = x.shape[0], self.point_num
        c0, c1, c2 = self.channels
        dev = x.device
        nxt = getattr(self, "_geom_next", None)
        object.__setattr__(self, "_geom_next", None)
        ready = getattr(self, "_geom_ready", None)
        object.__setattr__(self, "_geom_ready", None)
        if ready is not None:
            if ready["rows"][0] != B * P:
                raise ValueError("take_geometry(): the tables were built for another batch size")
            g = ready
        elif nxt is not None and nxt[1][1] == B:
            g = nxt[0].get(nxt[1])
            if g is None:
                g = nxt[0][nxt[1]] = self.geometry(x)
        else:
            g = self.geometry(x)
        R0, R1, R2 = g["rows"]
        e = lambda r, c: torch.empty(r, c, device=dev)
        vcat = self._vcat_ok()
        if vcat:
            m1, m0 = c2 // c1, c1 // c0
            comb1 = e(m1 * R2 + R1, c1)                         # [H2 as m1 rows of c1 per voxel | H1]
            comb0 = e(m0 * R1 + R0, c0)                         # [E1 as m0 rows of c0 per voxel | H0]
            H2, H1 = comb1[:m1 * R2].view(R2, c2), comb1[m1 * R2:]
            E1, H0 = comb0[:m0 * R1].view(R1, c1), comb0[m0 * R1:]
            cat0 = cat1 = None
        else:
            cat0 = e(R0, c1 + c0)                               # [unpool(E1) | H0]
            H0 = cat0[:, c1:]
            cat1 = e(R1, c2 + c1)                               # [unpool(H2) | H1]
            H1 = cat1[:, c2:]
            H2, E1 = e(R2, c2), e(R1, c1)
        cols0 = self._conv("conv0", g["feat0"], g["nbr0"], 4, H0, idx_pad=g["nbr0p"])
        D1 = e(R1, c1)
        colsd0 = self._conv("down0", H0, g["l1"]["child"], c0, D1)
        cols1 = self._conv("conv1", D1, g["nbr1"], c1, H1)
        D2 = e(R2, c2)
        colsd1 = self._conv("down1", H1, g["l2"]["child"], c1, D2)
        cols2 = self._conv("conv2", D2, g["nbr2"], c2, H2)
        E0 = e(R0, c0)
        if vcat:
            if "up1_idx" not in g:
                g["up1_idx"] = self._vcat_table(g["l2"]["parent"], m1, R2)
                g["up0_idx"] = self._vcat_table(g["l1"]["parent"], m0, R1)
            lin = self.up1
            ops.sparse_conv_fwd(comb1, g["up1_idx"], c1, lin.weight.data, lin.bias.data, E1, self._act, self._zero(dev))
            lin = self.up0
            ops.sparse_conv_fwd(comb0, g["up0_idx"], c0, lin.weight.data, lin.bias.data, E0, self._act, self._zero(dev))
        else:
            ops.rows_gather(H2, g["l2"]["parent"].view(-1, 1), c2, cat1[:, :c2])
            self._lin("up1", cat1, E1)
            ops.rows_gather(E1, g["l1"]["parent"].view(-1, 1), c1, cat0[:, :c1])
            self._lin("up0", cat0, E0)
        fbuf = torch.empty(B, c0 + self.proprio_shape, device=dev)
        arg = ops.maxpool_rows(E0, B, P, fbuf[:, :c0])
        if self.proprio_shape != 0:
            fbuf[:, c0:].copy_(x[:, -self.proprio_shape:])
        object.__setattr__(self, "_saved", dict(g=g, cols0=cols0, cat0=cat0, colsd0=colsd0, D1=D1, cols1=cols1, cat1=cat1,
                                                colsd1=colsd1, D2=D2, cols2=cols2, H2=H2, E0=E0, arg=arg, B=B, vcat=vcat,
                                                H0=H0, H1=H1, E1=E1, comb0=comb0 if vcat else None, comb1=comb1 if vcat else None))
        return self._head.forward(fbuf, out)

    def _decoder_backward_compact(self, s, g, dfbuf, ws):
        """Backward of max-pool, up0, up1 and conv2's weight gradient over the rows that carry gradient.

        The cloud-wide max-pool leaves ONE non-zero per (cloud, channel) in the gradient of E0, and everything above the coarsest
        convolution is row-local (1 x 1 layers on [un-pooled | skip] rows): the gradient is non-zero on at most c0 rows per cloud
        at level 0 (the winners), on their parents at level 1 and on their grand-parents at level 2 -- 32 of 4096 / ~1300 / ~300.
        Each level keeps the cloud's DISTINCT rows (c0 slots, ascending, padded: pm_rows_uniq_i32) and every row's slot; a coarse
        slot sums its children's gradient rows in slot order (pm_child_sum_f32: fixed order); the layers' GEMMs run on those
        B * c0 rows -- the same kernels, on compacted (gradient rows, table rows) pairs (pm_table_rows_i32: the padding slots
        read absent taps).  The skip halves' raw gradients stay COMPACT: the strided layers' backward picks them up through a
        row -> slot map (pm_rowmap_scatter_i32, pm_rows_gather_bwd_skip_f32) instead of a dense zero-filled tensor per level
        (1 GB + 0.7 GB of fills per backward at 2048 clouds).  Same sums as the dense form up to their order.  Every step is an
        entry point of include/partmanip_hip.h: nothing of this backward runs in the tensor library.

        (The dense form sums a coarse row's children through the `child` tables -- canonical rows only -- this one through
        `parent`, which every row has.  They agree because a duplicate-coordinate row can never be a max-pool winner: its table
        entries are its canonical twin's (its own centre tap points at the twin), so its E0 row EQUALS the twin's bit for bit,
        and the max-pool takes the lowest row among equals -- the twin, which by construction has the lower row index.)"""
        c0, c1, c2 = self.channels
        P, B = self.point_num, s["B"]
        dev = dfbuf.device
        R0, R1, R2 = g["rows"]
        W = lambda n: getattr(self, n).weight.data
        zero = self._zero(dev)
        e = lambda r, c: torch.empty(r, c, device=dev)
        N = B * c0
        u0, um0, rank0 = ops.rows_uniq(s["arg"], c0, R0, row_base=P)
        if _os.environ.get("PARTMANIP_DEBUG_COMPACT") == "1":
            # this backward sums a coarse row's children through `parent` (every row has one), the dense form through `child`
            # (canonical rows only) -- equal because a duplicate-coordinate row never wins the max-pool.  Debug check of exactly that
            # (a host read): every winner is the canonical row of its cell.
            win = u0[u0 < R0].long()
            pc = g["l1"]["parent_canon"].view(-1)[win]
            if bool((pc < 0).any()):
                raise RuntimeError("SparseUNet compact decoder backward: a duplicate-coordinate row won the max-pool")
        u1, um1, rank1 = ops.rows_uniq(u0, c0, R1, table=g["l1"]["parent"].view(-1), pad_in=R0)
        u2, um2, rank2 = ops.rows_uniq(u1, c0, R2, table=g["l2"]["parent"].view(-1), pad_in=R1)
        sel = lambda t, um: ops.rows_gather(t, um.view(N, 1), t.shape[1], e(N, t.shape[1]))      # padding slots: zero rows
        # level 0: pooled gradient -> the winners' rows (slot = rank of the channel's winner), times tanh'(E0)
        dzE0 = ops.maxpool_rows_bwd(dfbuf[:, :c0], rank0, c0, y_tanh=sel(s["E0"], um0))
        ops.sparse_conv_bwd_weight(dzE0, s["comb0"], ops.table_rows(g["up0_idx"], um0.view(-1)), c0, *self._g["up0"], zero, ws)
        skip0 = ops.linear_bwd_data(dzE0, W("up0")[:, c1:], None, e(N, c0), ops.ACT_NONE)              # skip half, raw, compact
        # level 1
        sum0 = ops.child_sum(dzE0, rank1, e(N, c0))
        dzE1 = ops.linear_bwd_data(sum0, W("up0")[:, :c1], sel(s["E1"], um1), e(N, c1), self._act)
        ops.sparse_conv_bwd_weight(dzE1, s["comb1"], ops.table_rows(g["up1_idx"], um1.view(-1)), c1, *self._g["up1"], zero, ws)
        skip1 = ops.linear_bwd_data(dzE1, W("up1")[:, c2:], None, e(N, c1), ops.ACT_NONE)
        # level 2
        sum1 = ops.child_sum(dzE1, rank2, e(N, c1))
        dzH2c = ops.linear_bwd_data(sum1, W("up1")[:, :c2], sel(s["H2"], um2), e(N, c2), self._act)
        ops.sparse_conv_bwd_weight(dzH2c, s["D2"], ops.table_rows(g["nbr2"], um2.view(-1)), c2, *self._g["conv2"], zero, ws)
        # conv2's data gradient: the column gradient of the B * c0 rows only ((rows x 27 c2) = dz W), then every level-2 row sums
        # the blocks of its neighbours THAT ARE among them (row -> compact row map; most taps miss) -- B * c0 x 27 c2 x c2 MACs
        # + a gather instead of the gathered GEMM over all of the level's rows
        dcols = ops.linear_bwd_data(dzH2c, W("conv2"), None, e(N, 27 * c2), ops.ACT_NONE)
        dzD2 = ops.rows_gather_bwd(dcols, g["nbr2"], c2, torch.empty_like(s["D2"]), reverse=True, self_col=13, y_tanh=s["D2"],
                                   rowmap=ops.rowmap_scatter(R2, u2.view(-1), R2))
        return (skip0, ops.rowmap_scatter(R0, u0.view(-1), R0)), (skip1, ops.rowmap_scatter(R1, u1.view(-1), R1)), dzD2

    def _hip_backward(self, dy):
        s, g = self._saved, self._saved["g"]
        c0, c1, c2 = self.channels
        P, B = self.point_num, s["B"]
        ws = self._workspace(dy.device)
        W = lambda n: getattr(self, n).weight.data
        dfbuf = torch.empty(B, c0 + self.proprio_shape, device=dy.device)
        self._head.backward(dy, ws, dx_out=dfbuf)
        H0, H1 = s["H0"], s["H1"]
        compact = self.sparse_top and s["vcat"] and s["cols2"] is None and c0 <= 64
        skip0 = skip1 = None
        if compact:
            skip0, skip1, dzD2 = self._decoder_backward_compact(s, g, dfbuf, ws)
            dzH0, dzH1 = torch.empty(g["rows"][0], c0, device=dy.device), torch.empty(g["rows"][1], c1, device=dy.device)
            acc_mode = False
        else:
            dzE0 = ops.maxpool_rows_bwd(dfbuf[:, :c0], s["arg"], P, y_tanh=s["E0"])        # pre-activation gradient of up0
        if compact:
            pass
        elif s["vcat"]:
            # the concatenated operands are virtual (see _vcat_ok): weight gradients gather them again; the data gradients are
            # taken RAW (no activation derivative: nothing to read back) -- tanh' of the un-pooled half is applied after the sum
            # over a coarse row's children, tanh' of the skip half after the strided layer's contribution has been added
            # The un-pooled half by linearity: sum_children (dz[child] W_u) = (sum_children dz[child]) W_u -- the children's
            # gradient rows are summed FIRST (c_out-wide rows, fixed order) and the GEMM runs on the coarse level's rows, so
            # the (fine rows x c_hi) block of the data gradient is never formed.
            zero = self._zero(dy.device)
            e = lambda r, c: torch.empty(r, c, device=dy.device)
            ops.sparse_conv_bwd_weight(dzE0, s["comb0"], g["up0_idx"], c0, *self._g["up0"], zero, ws)
            sum0 = ops.rows_gather_bwd(dzE0, g["l1"]["child"], c0, e(s["D1"].shape[0], c0), mode=2)
            dzE1 = ops.linear_bwd_data(sum0, W("up0")[:, :c1], s["E1"], e(s["D1"].shape[0], c1), self._act)
            dzH0 = ops.linear_bwd_data(dzE0, W("up0")[:, c1:], None, e(dzE0.shape[0], c0), ops.ACT_NONE)      # skip half, raw
            ops.sparse_conv_bwd_weight(dzE1, s["comb1"], g["up1_idx"], c1, *self._g["up1"], zero, ws)
            sum1 = ops.rows_gather_bwd(dzE1, g["l2"]["child"], c1, e(s["D2"].shape[0], c1), mode=2)
            dzH2 = ops.linear_bwd_data(sum1, W("up1")[:, :c2], s["H2"], e(s["D2"].shape[0], c2), self._act)
            dzH1 = ops.linear_bwd_data(dzE1, W("up1")[:, c2:], None, e(dzE1.shape[0], c1), ops.ACT_NONE)
            acc_mode = 2                                                                    # (skip + strided) * tanh'
        else:
            cat0, cat1 = s["cat0"], s["cat1"]
            ops.linear_bwd_weight(dzE0, cat0, *self._g["up0"], ws)
            dcat0 = torch.empty_like(cat0)
            ops.linear_bwd_data(dzE0, W("up0"), cat0, dcat0, self._act)                    # tanh' of both halves folded in
            dzE1 = ops.rows_gather_bwd(dcat0[:, :c1], g["l1"]["child"], c1, torch.empty_like(s["D1"]), mode=2)
            ops.linear_bwd_weight(dzE1, cat1, *self._g["up1"], ws)
            dcat1 = torch.empty_like(cat1)
            ops.linear_bwd_data(dzE1, W("up1"), cat1, dcat1, self._act)
            dzH2 = ops.rows_gather_bwd(dcat1[:, :c2], g["l2"]["child"], c2, torch.empty_like(s["D2"]), mode=2)
            dzH1, dzH0 = dcat1[:, c2:], dcat0[:, c1:]
            acc_mode = True                                                                 # skip part already pre-activation
        if not compact:
            self._conv_wgrad("conv2", dzH2, s["D2"], g["nbr2"], c2, s["cols2"], ws)
            dzD2 = self._conv_dgrad("conv2", dzH2, g["nbr2"], s["D2"], torch.empty_like(s["D2"]))
        self._conv_wgrad("down1", dzD2, H1, g["l2"]["child"], c1, s["colsd1"], ws)
        dcolsd1 = torch.empty(dzD2.shape[0], 8 * c1, device=dy.device)
        ops.linear_bwd_data(dzD2, W("down1"), None, dcolsd1, ops.ACT_NONE)
        ops.rows_gather_bwd(dcolsd1, g["l2"]["parent_canon"].view(-1, 1), c1, dzH1, tslot=g["l2"]["slot"].view(-1, 1), mode=1,
                            y_tanh=H1, accumulate=acc_mode, skip=skip1)
        self._conv_wgrad("conv1", dzH1, s["D1"], g["nbr1"], c1, s["cols1"], ws)
        dzD1 = self._conv_dgrad("conv1", dzH1, g["nbr1"], s["D1"], torch.empty_like(s["D1"]))
        self._conv_wgrad("down0", dzD1, H0, g["l1"]["child"], c0, s["colsd0"], ws)
        dcolsd0 = torch.empty(dzD1.shape[0], 8 * c0, device=dy.device)
        ops.linear_bwd_data(dzD1, W("down0"), None, dcolsd0, ops.ACT_NONE)
        ops.rows_gather_bwd(dcolsd0, g["l1"]["parent_canon"].view(-1, 1), c0, dzH0, tslot=g["l1"]["slot"].view(-1, 1), mode=1,
                            y_tanh=H0, accumulate=acc_mode, skip=skip0)
        self._conv_wgrad("conv0", dzH0, g["feat0"], g["nbr0"], 4, s["cols0"], ws)           # the input features are data


class _ConvEncoder(nn.Module):
    """Parameter container with the reference's names (network.py:116-139 `Encoder`: conv1..conv3)."""

    def __init__(self, in_channels, filters, kernels, strides):
        super().__init__()
        chans = [in_channels] + list(filters)
        for i in range(3):
            setattr(self, f"conv{i + 1}", nn.Conv3d(chans[i], chans[i + 1], kernels[i], stride=strides[i],
                                                    padding=kernels[i] // 2))


class Conv3DNet(_HipNet):
    """network.py:67-94: TSDF student.  Encoder = Conv3d(1,16,k5,s3) - act - Conv3d(16,32,k3,s3) - act -
    Conv3d(32,32,k3,s2) - act on a res^3 volume (50^3 -> 17^3 -> 6^3 -> 3^3), flatten (channels first, 32*27)
    [+ proprio] -> Linear 256 - act - Linear out.  `state_dict` keys and default initialisation are the
    reference's (`encoder.conv{1,2,3}.{weight,bias}`, `final_mlp.{0,2}.*`).
    The single-channel input layer is a direct 125-tap stencil (pm_conv3d_c1_fwd_f32 / pm_conv3d_c1_wgrad_f32: no patch
    matrix); layers 2-3 run as a patch gather (pm_im2col3d_f32) + the fp32 MFMA Linear kernel on conv.weight viewed
    (Cout, Cin*k^3); layer outputs stay channels-last ((b, d, h, w) rows x Cout) and are read through strides by the
    next gather.  Backward: Linear backward kernels + pm_col2im3d_f32 (which folds tanh')."""

    FILTERS, KERNELS, STRIDES = (16, 32, 32), (5, 3, 3), (3, 3, 2)

    def __init__(self, input_dim, output_dim, net_cfg, proprio_shape):
        super().__init__()
        self.res = round(input_dim ** (1 / 3))
        act = net_cfg['activation']
        # every activation of get_activation (network.py:7-24, :70): the layers are GEMMs with activation epilogues; the
        # single-channel input layer's direct stencil kernels are tanh kernels, other activations take its patch-matrix form
        code = _act_code(act, linear_only=True)
        self.encoder = _ConvEncoder(1, self.FILTERS, self.KERNELS, self.STRIDES)
        self.activation = get_activation(act)
        self.final_mlp = nn.Sequential(nn.Linear(32 * 27 + proprio_shape, 256), self.activation, nn.Linear(256, output_dim))
        self.proprio_shape = proprio_shape
        ext = [self.res]
        for k, st in zip(self.KERNELS, self.STRIDES):
            ext.append(ops.conv3d_out(ext[-1], k, st, k // 2))
        if ext[-1] ** 3 * self.FILTERS[-1] != 32 * 27:
            raise ValueError(f"Conv3DNet: a {self.res}^3 volume does not reduce to 3^3 (network.py:75 hard-codes 32*27)")
        object.__setattr__(self, "_ext", ext)
        object.__setattr__(self, "_act", code)
        object.__setattr__(self, "_head", _LinearChain([self.final_mlp[0], self.final_mlp[2]], code))
        object.__setattr__(self, "_conv_grads", None)
        object.__setattr__(self, "_w1p", None)
        self.fused_gather = bool(net_cfg.get('fused_gather', True))         # False: im2col + Linear for layers 2-3 (A/B)

    def _convs(self):
        return [self.encoder.conv1, self.encoder.conv2, self.encoder.conv3]

    def _zero(self, device):
        z = getattr(self, "_zero_row", None)
        if z is None or z.device != device:
            z = torch.zeros(128, device=device)
            object.__setattr__(self, "_zero_row", z)
        return z

    def _patch_table(self, B, n, k, stride, pad, cin, device):
        """(B*no^3, J) int32: row of the channels-last input (b, d, h, w) under tap (kd, kh, kw) of output (b, od, oh, ow), -1
        in the padding; J = k^3 rounded up so that J*cin is whole K-steps of 32 (extra taps are -1).  Depends on the batch
        size and the geometry only: cached."""
        key = (B, n, k, stride, pad, cin, str(device))
        cache = getattr(self, "_tables", None)
        if cache is None:
            cache = {}
            object.__setattr__(self, "_tables", cache)
        if key not in cache:
            no = ops.conv3d_out(n, k, stride, pad)
            o = torch.arange(no, device=device) * stride - pad
            t = torch.arange(k, device=device)
            p = o[:, None] + t[None, :]                                       # (no, k) input coordinate per (output, tap)
            ok = (p >= 0) & (p < n)
            d, h, w = p[:, None, None, :, None, None], p[None, :, None, None, :, None], p[None, None, :, None, None, :]
            okk = ok[:, None, None, :, None, None] & ok[None, :, None, None, :, None] & ok[None, None, :, None, None, :]
            lin = ((d * n + h) * n + w).expand(no, no, no, k, k, k)
            lin = torch.where(okk, lin, torch.full_like(lin, -1)).reshape(no ** 3, k ** 3)
            b = torch.arange(B, device=device)[:, None, None] * n ** 3
            idx = torch.where(lin[None] >= 0, lin[None] + b, lin[None]).reshape(B * no ** 3, k ** 3)
            jp = k ** 3
            while (jp * cin) % 32:
                jp += 1
            if jp > k ** 3:
                idx = torch.nn.functional.pad(idx, (0, jp - k ** 3), value=-1)
            cache[key] = (idx.to(torch.int32).contiguous(), jp)
        return cache[key]

    @staticmethod
    def _tap_major(conv, jp):
        """conv.weight (Cout, Cin, k, k, k) as (Cout, jp*Cin) with columns (tap, c), zero columns for the padding taps."""
        co, ci = conv.out_channels, conv.in_channels
        k3 = conv.weight[0, 0].numel()
        wp = torch.zeros(co, jp * ci, device=conv.weight.device)
        wp[:, :k3 * ci].view(co, k3, ci).copy_(conv.weight.data.view(co, ci, k3).transpose(1, 2))
        return wp

    def set_grad_views(self, views):
        self._head.grads = [(views[f"final_mlp.{i}.weight"], views[f"final_mlp.{i}.bias"]) for i in (0, 2)]
        object.__setattr__(self, "_conv_grads", [(views[f"encoder.conv{i}.weight"], views[f"encoder.conv{i}.bias"])
                                                 for i in (1, 2, 3)])

    supports_row_index = True        # hip_forward(store, rows=idx): the batch is rows idx of `store`, read in place

    def hip_forward(self, x, out=None, rows=None):
        """rows (int64 device tensor, optional): the mini-batch is x[rows] -- DAgger draws random rows of its ring -- and the
        input layer's kernels read those volumes where they lie (no 0.8 GB gathered copy per 1600-row mini-batch)."""
        r = self.res
        if rows is not None and not (self._act in (ops.ACT_NONE, ops.ACT_TANH) and r <= 64
                                     and ops.conv3d_c1_supported(self.KERNELS[0], self.FILTERS[0])):
            x, rows = x.index_select(0, rows), None
        B = x.shape[0] if rows is None else rows.numel()
        vol = x[:, :r ** 3].unflatten(1, (1, r, r, r))                   # 5-D view (row stride = the obs width)
        convs, ext = self._convs(), self._ext
        # conv1's K = 125 is not a multiple of 4: a (16,128) zero-padded copy of the weight lets the GEMM use 16-B loads
        w1 = convs[0].weight.data.view(self.FILTERS[0], -1)
        k1 = w1.shape[1]
        k1p = (k1 + 3) // 4 * 4
        if self._w1p is None or self._w1p.device != x.device:
            object.__setattr__(self, "_w1p", torch.zeros(self.FILTERS[0], k1p, device=x.device))
        self._w1p[:, :k1].copy_(w1)
        saved = []
        cur = vol
        for i, conv in enumerate(convs):
            k, st = self.KERNELS[i], self.STRIDES[i]
            e = ext[i + 1]
            if (i == 0 and self._act in (ops.ACT_NONE, ops.ACT_TANH) and self.res <= 64
                    and ops.conv3d_c1_supported(k, conv.out_channels)):
                # the single-channel input layer runs as a direct stencil: no 4 GB patch matrix (csrc/conv3d.hip)
                y = ops.conv3d_c1_fwd(cur, k, st, k // 2, w1.t().contiguous(), conv.bias.data, self._act, rows)
                saved.append((cur, None, y))
                cur = y.view(B, e, e, e, conv.out_channels).permute(0, 4, 1, 2, 3)
                continue
            if i > 0 and self.fused_gather and conv.in_channels % 4 == 0:
                # layers 2-3 read channels-last rows of the previous layer: the patch gather runs inside the GEMM's LDS-DMA
                # loader (the same gathered operand as the sparse convolutions), no patch matrix in HBM
                idx, jp = self._patch_table(B, ext[i], k, st, k // 2, conv.in_channels, x.device)
                wp = self._tap_major(conv, jp)
                src = saved[i - 1][2]
                y = torch.empty(idx.shape[0], conv.out_channels, device=x.device)
                ops.sparse_conv_fwd(src, idx, conv.in_channels, wp, conv.bias.data, y, self._act, self._zero(x.device))
                saved.append((cur, idx, y))
                cur = y.view(B, e, e, e, conv.out_channels).permute(0, 4, 1, 2, 3)
                continue
            w = self._w1p if i == 0 else conv.weight.data.view(conv.out_channels, -1)
            cols = ops.im2col3d(cur, k, st, k // 2, w.shape[1])
            y = torch.empty(cols.shape[0], conv.out_channels, device=x.device)
            ops.linear_fwd(cols, w, conv.bias.data, y, self._act)
            saved.append((cur, cols, y))
            cur = y.view(B, e, e, e, conv.out_channels).permute(0, 4, 1, 2, 3)      # channels-last storage, NCDHW view
        fbuf = torch.empty(B, 32 * 27 + self.proprio_shape, device=x.device)
        fbuf[:, :32 * 27].view(B, 32, 27).copy_(cur.reshape(B, 32, 27))           # channels-first flatten (network.py:88,90)
        if self.proprio_shape != 0:
            tail = x[:, -self.proprio_shape:]
            fbuf[:, 32 * 27:].copy_(tail if rows is None else tail.index_select(0, rows))
        object.__setattr__(self, "_saved", saved)
        object.__setattr__(self, "_rows_idx", rows)
        return self._head.forward(fbuf, out)

    def hip_backward(self, dy):
        saved, B = self._saved, dy.shape[0]
        ws = self._workspace(dy.device)
        convs, ext = self._convs(), self._ext
        dfbuf = torch.empty(B, 32 * 27 + self.proprio_shape, device=dy.device)
        self._head.backward(dy, ws, dx_out=dfbuf, x_is_activation=True)       # d z3 in the flattened layout (tanh' folded)
        dz = torch.empty(B * 27, 32, device=dy.device)
        dz.view(B, 27, 32).copy_(dfbuf[:, :32 * 27].view(B, 32, 27).transpose(1, 2))
        for i in (2, 1, 0):
            conv, (x5, cols, y) = convs[i], saved[i]
            k, st = self.KERNELS[i], self.STRIDES[i]
            dW, db = self._conv_grads[i]
            if i == 0 and cols is None:
                ops.conv3d_c1_wgrad(dz, x5, k, st, k // 2, dW.view(conv.out_channels, -1), db, ws, self._rows_idx)
                break
            if i == 0:
                dwp = torch.empty_like(self._w1p)
                ops.linear_bwd_weight(dz, cols, dwp, db, ws)
                dW.view(conv.out_channels, -1).copy_(dwp[:, :dW[0].numel()])
                break                                                        # the volume is data: no gradient to it
            if cols.dtype == torch.int32:                                    # fused layer: `cols` is its patch table
                cin, k3 = conv.in_channels, k ** 3
                dwp = torch.empty(conv.out_channels, cols.shape[1] * cin, device=dy.device)
                ops.sparse_conv_bwd_weight(dz, saved[i - 1][2], cols, cin, dwp, db, self._zero(dy.device), ws)
                dW.view(conv.out_channels, cin, k3).copy_(dwp[:, :k3 * cin].view(conv.out_channels, k3, cin).transpose(1, 2))
                if st == k:
                    # stride == k: the patches do not overlap, every input voxel is idx[r, j] for at most one (r, j): the data
                    # gradient is one GEMM whose epilogue scatters to the input rows -- no column gradients, no col2im
                    y_prev = saved[i - 1][2]
                    n_in, no = ext[i], ext[i + 1]
                    covered = no * k >= n_in + k // 2
                    dzp = torch.empty_like(y_prev) if covered else torch.zeros_like(y_prev)
                    ops.sparse_conv_bwd_data_scatter(dz, self._tap_major(conv, cols.shape[1]), cols, cin, y_prev, dzp, self._act)
                    dz = dzp
                    continue
                dcols = torch.empty(cols.shape[0], cin * k3, device=dy.device)
            else:
                ops.linear_bwd_weight(dz, cols, dW.view(conv.out_channels, -1), db, ws)
                dcols = torch.empty_like(cols)
            ops.linear_bwd_data(dz, conv.weight.data.view(conv.out_channels, -1), None, dcols, ops.ACT_NONE)
            y_prev = saved[i - 1][2]                                          # this layer's input = previous tanh output
            dz = torch.empty_like(y_prev)
            e, c = ext[i], convs[i - 1].out_channels
            as5 = lambda t: t.view(B, e, e, e, c).permute(0, 4, 1, 2, 3)
            ops.col2im3d(dcols, as5(dz), k, st, k // 2, as5(y_prev), self._act)


def _out_of_scope(name):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(
                f"backbone '{name}' (reference network.py) is an image / pooled-TSDF student outside this build's hot-path "
                "scope (SURVEY.md §8f rank 4); MLP, PointNet, PointNet2 and Conv3DNet are implemented")
    _Stub.__name__ = name
    return _Stub


PoolConv3DNet = _out_of_scope("PoolConv3DNet")
ResNet = _out_of_scope("ResNet")
depthResNet = _out_of_scope("depthResNet")
