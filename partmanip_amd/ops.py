"""Tensor-level wrappers over the C ABI (include/partmanip_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every op below passes
raw device pointers to libpartmanip_hip.so.  Ops refuse CPU tensors -- there is no CPU path.
"""
import ctypes as C
import os

import torch

from ._lib import lib, check

ACT_NONE, ACT_TANH, ACT_RELU, ACT_LRELU, ACT_ELU, ACT_SELU, ACT_SIGMOID = range(7)      # PM_ACT_* of the header


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class KernelTimer:
    """HIP-event bracket around named launches (bench.py's live roofline measurement).
    Events are recorded on torch's current stream, which is the stream every op launches on."""

    def __init__(self):
        self.enabled = set()
        self.events = {}

    def enable(self, *names):
        self.enabled = set(names)
        self.events = {n: [] for n in names}

    def add(self, *names):
        """Enable more names without dropping the events collected so far."""
        self.enabled |= set(names)
        for n in names:
            self.events.setdefault(n, [])

    def disable(self):
        self.enabled = set()

    def bracket(self, name):
        return _Bracket(self, name) if name in self.enabled else _NULL

    def mean_ms(self, name):
        ev = self.events.get(name, [])
        if not ev:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev)


class _Bracket:
    def __init__(self, timer, name):
        self.t, self.n = timer, name

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.t.events[self.n].append((self.a, self.b))


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _Null()
TIMER = KernelTimer()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("partmanip_amd ops run on MI355X only: got a CPU tensor (no CPU fallback exists)")


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")


def _rows(t, name):
    """(ptr-able 2-D view, row stride in elements); last dim must be contiguous."""
    if t.dim() != 2 or t.dtype != torch.float32 or (t.shape[1] != 1 and t.stride(1) != 1):   # a size-1 inner dim may carry any stride
        raise ValueError(f"{name}: expected a 2-D float32 tensor with unit inner stride, got {tuple(t.shape)} {t.stride()}")
    return t.stride(0)


class Workspace:
    """Grow-only device scratch buffer handed to the C ABI (the library never allocates)."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self.buf


# ----------------------------------------------------------------------------- K1-K3
def gae_scan(rewards, values, dones, succs, last_values, returns, advantages, gamma, lam, succ_value):
    _req(rewards, values, dones, succs, last_values, returns, advantages)
    T, N = rewards.shape[0], rewards.shape[1]
    for t, n in ((rewards, "rewards"), (values, "values"), (returns, "returns"), (advantages, "advantages"),
                 (last_values, "last_values")):
        _f32c(t, n)
    d8 = dones.view(torch.uint8) if dones.dtype == torch.bool else dones
    s8 = succs.view(torch.uint8) if succs.dtype == torch.bool else succs
    use = succ_value is not None
    with TIMER.bracket("gae_scan"):
        check(lib.pm_gae_scan_f32(_ptr(rewards), _ptr(values), _ptr(d8), _ptr(s8), _ptr(last_values), _ptr(returns),
                                  _ptr(advantages), T, N, float(gamma), float(gamma * lam), int(use),
                                  float(succ_value) if use else 0.0, _stream()), "pm_gae_scan_f32")


def moments(x, out2, ws):
    _req(x, out2)
    _f32c(x, "x")
    n = x.numel()
    w = ws.get(lib.pm_moments_workspace_bytes(n))
    check(lib.pm_moments_f64(_ptr(x), n, _ptr(out2), _ptr(w), w.numel(), _stream()), "pm_moments_f64")


def normalize_apply(x, mom2, count, eps=1e-8):
    _req(x, mom2)
    _f32c(x, "x")
    check(lib.pm_normalize_apply_f32(_ptr(x), x.numel(), _ptr(mom2), float(count), float(eps), _stream()),
          "pm_normalize_apply_f32")


def gather_rows(src2d, idx, dst2d):
    _req(src2d, idx, dst2d)
    ls, ld = _rows(src2d, "src"), _rows(dst2d, "dst")
    if idx.dtype != torch.int64:
        raise ValueError("idx must be int64")
    check(lib.pm_gather_rows_f32(_ptr(src2d), _ptr(idx), _ptr(dst2d), idx.numel(), src2d.shape[1], ls, ld, _stream()),
          "pm_gather_rows_f32")


# ----------------------------------------------------------------------------- K4/K5
def linear_fwd(x, w, b, y, act):
    _req(x, w, b, y)
    M, K = x.shape
    N = w.shape[0]
    with (TIMER.bracket(f"linear_fwd_{M}x{K}x{N}") if TIMER.enabled else _NULL):     # (bench.py names the glue GEMMs it wants timed by shape)
        check(lib.pm_linear_fwd_f32(_ptr(x), _rows(x, "x"), _ptr(w), _rows(w, "w"), _ptr(b), _ptr(y), _rows(y, "y"),
                                    M, N, K, act, _stream()), "pm_linear_fwd_f32")


def linear_bwd_data(dy, w, h, dx, act):
    _req(dy, w, h, dx)
    M, N = dy.shape
    K = w.shape[1]
    with (TIMER.bracket(f"linear_bwd_data_{M}x{N}x{K}") if TIMER.enabled else _NULL):
        check(lib.pm_linear_bwd_data_f32(_ptr(dy), _rows(dy, "dy"), _ptr(w), _rows(w, "w"), _ptr(h),
                                         _rows(h, "h") if h is not None else 0, _ptr(dx), _rows(dx, "dx"), M, N, K, act,
                                         _stream()), "pm_linear_bwd_data_f32")
    return dx


def linear_bwd_weight(dy, x, dw, db, ws):
    _req(dy, x, dw, db)
    M, N = dy.shape
    K = x.shape[1]
    w = ws.get(lib.pm_linear_bwd_weight_workspace_bytes(M, N, K))
    with (TIMER.bracket(f"linear_bwd_weight_{M}x{N}x{K}") if TIMER.enabled else _NULL):
        check(lib.pm_linear_bwd_weight_f32(_ptr(dy), _rows(dy, "dy"), _ptr(x), _rows(x, "x"), _ptr(dw), _rows(dw, "dw"),
                                           _ptr(db), M, N, K, _ptr(w), w.numel(), _stream()), "pm_linear_bwd_weight_f32")


# ---- grouped forms: several independent problems of one kind in ONE launch (the actor's and the critic's layer l; all
# weight gradients of a backward pass).  Each item is the argument tuple of the single-problem op.
def linear_fwd_group(items):
    """items: [(x, w, b, y, act), ...]"""
    from ._lib import LinearFwdDesc
    arr = (LinearFwdDesc * len(items))()
    for d, (x, w, b, y, act) in zip(arr, items):
        _req(x, w, b, y)
        d.X, d.ldx, d.W, d.ldw, d.b, d.Y, d.ldy = _ptr(x), _rows(x, "x"), _ptr(w), _rows(w, "w"), _ptr(b), _ptr(y), _rows(y, "y")
        d.M, d.K, d.N, d.act = x.shape[0], x.shape[1], w.shape[0], act
    check(lib.pm_linear_fwd_group_f32(len(items), arr, _stream()), "pm_linear_fwd_group_f32")


def linear_bwd_data_group(items):
    """items: [(dy, w, h, dx, act), ...]"""
    from ._lib import LinearBwdDataDesc
    arr = (LinearBwdDataDesc * len(items))()
    for d, (dy, w, h, dx, act) in zip(arr, items):
        _req(dy, w, h, dx)
        d.dY, d.lddy, d.W, d.ldw, d.dX, d.lddx = _ptr(dy), _rows(dy, "dy"), _ptr(w), _rows(w, "w"), _ptr(dx), _rows(dx, "dx")
        d.H, d.ldh = _ptr(h), (_rows(h, "h") if h is not None else 0)
        d.M, d.N, d.K, d.act = dy.shape[0], dy.shape[1], w.shape[1], act
    check(lib.pm_linear_bwd_data_group_f32(len(items), arr, _stream()), "pm_linear_bwd_data_group_f32")


PM_EUNSUPPORTED = -4


def _chain_ws(ws, kind, M, N, n_layers):
    """The monotonic stripe counters + error word of one chain shape (zeroed once, never shared between shapes / directions)."""
    n = int(lib.pm_linear_chain_workspace_bytes(M)) // 8
    tab = getattr(ws, "chain_ctr", None)
    if tab is None:
        tab = ws.chain_ctr = {}
    key = (kind, M, N, n_layers)
    if key not in tab:
        tab[key] = torch.zeros(n, dtype=torch.int64, device=ws.device)
    return tab[key], n


def linear_fwd_chain(items, ws):
    """items: [(x, w, b, y, act), ...] with items[i + 1].x is items[i].y: consecutive layers in ONE launch (pm_linear_fwd_chain_f32).
    Returns False when the shapes do not fit the scheme (nothing was launched: issue the layers one by one)."""
    from ._lib import LinearFwdDesc
    arr = (LinearFwdDesc * len(items))()
    for d, (x, w, b, y, act) in zip(arr, items):
        _req(x, w, b, y)
        d.X, d.ldx, d.W, d.ldw, d.b, d.Y, d.ldy = _ptr(x), _rows(x, "x"), _ptr(w), _rows(w, "w"), _ptr(b), _ptr(y), _rows(y, "y")
        d.M, d.K, d.N, d.act = x.shape[0], x.shape[1], w.shape[0], act
    ctr, n = _chain_ws(ws, "fwd", items[0][0].shape[0], items[0][1].shape[0], len(items))
    rc = lib.pm_linear_fwd_chain_f32(len(items), arr, _ptr(ctr), n * 8, _stream())
    if rc == PM_EUNSUPPORTED:
        return False
    check(rc, "pm_linear_fwd_chain_f32")
    return True


def linear_bwd_data_chain(items, ws):
    """items: [(dy, w, h, dx, act), ...] top layer first, items[i + 1].dy is items[i].dx (pm_linear_bwd_data_chain_f32)."""
    from ._lib import LinearBwdDataDesc
    arr = (LinearBwdDataDesc * len(items))()
    for d, (dy, w, h, dx, act) in zip(arr, items):
        _req(dy, w, h, dx)
        d.dY, d.lddy, d.W, d.ldw, d.dX, d.lddx = _ptr(dy), _rows(dy, "dy"), _ptr(w), _rows(w, "w"), _ptr(dx), _rows(dx, "dx")
        d.H, d.ldh = _ptr(h), (_rows(h, "h") if h is not None else 0)
        d.M, d.N, d.K, d.act = dy.shape[0], dy.shape[1], w.shape[1], act
    ctr, n = _chain_ws(ws, "bwd", items[0][0].shape[0], items[0][1].shape[1], len(items))
    rc = lib.pm_linear_bwd_data_chain_f32(len(items), arr, _ptr(ctr), n * 8, _stream())
    if rc == PM_EUNSUPPORTED:
        return False
    check(rc, "pm_linear_bwd_data_chain_f32")
    return True


def chain_gave_up(ws):
    """True when a stripe barrier of the last chain launch on this workspace ran into its spin limit (host sync)."""
    tab = getattr(ws, "chain_ctr", None) or {}
    bad = [t for t in tab.values() if int(t[-1].item()) != 0]
    for t in bad:                # counters are monotonic across launches: after a failed launch they are off a launch boundary --
        t.zero_()                # start over (error word included), so that the NEXT launch is valid again
    return bool(bad)


def padded_cols(rows, cols, device, zero=False):
    """(rows, cols) fp32 view of a buffer whose rows are padded to a multiple of 4 floats; the view carries `_pm_cols` (the
    readable row width), which `linear_bwd_weight_group` hands to the kernels so that a 10- or 53-column operand takes the
    16-byte loaders."""
    w = (cols + 3) // 4 * 4
    buf = (torch.zeros if zero else torch.empty)(rows, w, device=device)
    v = buf[:, :cols]
    v._pm_cols = w
    return v


def linear_bwd_weight_group(items, splits=1):
    """items: [(dy, x, dw, db, slab_stride), ...]; splits > 1: slab z of dw / db lands at + z * slab_stride elements and the
    caller sums the slabs (clip_adam_group does)."""
    from ._lib import LinearBwdWeightDesc
    arr = (LinearBwdWeightDesc * len(items))()
    for d, (dy, x, dw, db, stride) in zip(arr, items):
        _req(dy, x, dw, db)
        # tensors whose allocator marked their rows as readable past the last column (`padded_cols`): 16-byte loaders
        d.dy_cols, d.x_cols = int(getattr(dy, "_pm_cols", 0)), int(getattr(x, "_pm_cols", 0))
        d.dY, d.lddy, d.X, d.ldx, d.dW, d.lddw, d.db = _ptr(dy), _rows(dy, "dy"), _ptr(x), _rows(x, "x"), _ptr(dw), _rows(dw, "dw"), _ptr(db)
        d.slab_stride, d.M, d.N, d.K = int(stride), dy.shape[0], dy.shape[1], x.shape[1]
    check(lib.pm_linear_bwd_weight_group_f32(len(items), arr, int(splits), _stream()), "pm_linear_bwd_weight_group_f32")


def clip_adam_group(items):
    """items: [dict(p, g, m, v, n_clip, max_norm, lr, b1, b2, eps, state, skip_flag, gnorm, ws, extra, extra_stride, n_sum,
    n_extra)]: pm_clip_adam_step_f32 for several optimisers in two launches; `extra` = split-K slabs 1.. of g[:n_sum]."""
    from ._lib import ClipAdamDesc
    arr = (ClipAdamDesc * len(items))()
    for d, it in zip(arr, items):
        p, g = it["p"], it["g"]
        _req(p, g, it["m"], it["v"], it["state"], it.get("skip_flag"), it.get("gnorm"), it.get("extra"))
        n = p.numel()
        w = it["ws"].get(lib.pm_clip_adam_workspace_bytes(n) + 8)
        base = w.data_ptr()
        d.params, d.grads, d.exp_avg, d.exp_avg_sq, d.n, d.n_clip = _ptr(p), _ptr(g), _ptr(it["m"]), _ptr(it["v"]), n, int(it["n_clip"])
        d.extra, d.extra_stride, d.n_sum, d.n_extra = _ptr(it.get("extra")), int(it.get("extra_stride", 0)), int(it.get("n_sum", 0)), int(it.get("n_extra", 0))
        d.max_norm, d.lr, d.b1, d.b2, d.eps = float(it["max_norm"]), float(it["lr"]), float(it["b1"]), float(it["b2"]), float(it["eps"])
        d.state, d.skip_flag, d.gnorm_out, d.workspace = _ptr(it["state"]), _ptr(it.get("skip_flag")), _ptr(it.get("gnorm")), base + (-base) % 8
        dp = it.get("dp")                                  # (1/W, scal or None, desired_kl or 0): data-parallel step, see the header
        if dp is not None:
            _req(dp[1])
            d.grad_scale, d.dp_scal, d.dp_kl_desired = float(dp[0]), _ptr(dp[1]), float(dp[2])
        st = it.get("stats")                               # (acc, scal, which): ppo_accumulate_stats inside the norm pass
        if st is not None:
            _req(st[0], st[1])
            d.stats_acc, d.stats_scal, d.stats_which = _ptr(st[0]), _ptr(st[1]), int(st[2])
    check(lib.pm_clip_adam_group_f32(len(items), arr, _stream()), "pm_clip_adam_group_f32")


def grad_slab_sum(g, extra, extra_stride, n_sum, n_extra):
    """g[:n_sum] += the n_extra split-K slabs (pm_grad_slab_sum_f32): before a gradient all-reduce."""
    _req(g, extra)
    check(lib.pm_grad_slab_sum_f32(_ptr(g), _ptr(extra), int(extra_stride), int(n_sum), int(n_extra), _stream()), "pm_grad_slab_sum_f32")


# ----------------------------------------------------------------------------- K6/K7
def pointnet_packed_elems():
    return int(lib.pm_pointnet_packed_elems())


def pointnet_pack(w2, w3, packed):
    _req(w2, w3, packed)
    check(lib.pm_pointnet_pack_weights_f32(_ptr(w2), _ptr(w3), _ptr(packed), _stream()), "pm_pointnet_pack_weights_f32")


def pointnet_enc_fwd(x, P, Cc, sub_mean, w1, b1, b2, b3, packed, max_mean, feat, argmax, h2_save=None, act=None):
    """h2_save: optional (B, P, 256) buffer the training forward fills with the layer-2 activations; act: ACT_* of the
    two hidden layers (default tanh)."""
    _req(x, w1, b1, b2, b3, packed, feat, argmax, h2_save)
    B = x.shape[0]
    with TIMER.bracket("pointnet_enc_fwd"):
        check(lib.pm_pointnet_enc_fwd_f32(_ptr(x), _rows(x, "x"), B, P, Cc, int(sub_mean), _ptr(w1), _ptr(b1), _ptr(b2),
                                          _ptr(b3), _ptr(packed), int(max_mean), _ptr(feat), _rows(feat, "feat"),
                                          _ptr(argmax), _ptr(h2_save), int(ACT_TANH if act is None else act), _stream()),
              "pm_pointnet_enc_fwd_f32")


def pointnet_pack_bf3(w2, w3, packed):
    _req(w2, w3, packed)
    check(lib.pm_pointnet_pack_weights_bf3(_ptr(w2), _ptr(w3), _ptr(packed), _stream()), "pm_pointnet_pack_weights_bf3")


def pointnet_enc_fwd_bf3(x, P, Cc, sub_mean, w1, b1, b2, b3, packed, max_mean, feat, argmax, h2_save=None):
    _req(x, w1, b1, b2, b3, packed, feat, argmax)
    B = x.shape[0]
    with TIMER.bracket("pointnet_enc_fwd"):
        check(lib.pm_pointnet_enc_fwd_bf3(_ptr(x), _rows(x, "x"), B, P, Cc, int(sub_mean), _ptr(w1), _ptr(b1), _ptr(b2),
                                          _ptr(b3), _ptr(packed), int(max_mean), _ptr(feat), _rows(feat, "feat"),
                                          _ptr(argmax), _ptr(h2_save), _stream()), "pm_pointnet_enc_fwd_bf3")


def pointnet_pack_bf6(w2, w3, packed):
    _req(w2, w3, packed)
    check(lib.pm_pointnet_pack_weights_bf6(_ptr(w2), _ptr(w3), _ptr(packed), _stream()), "pm_pointnet_pack_weights_bf6")


def pointnet_enc_fwd_bf6(x, P, Cc, sub_mean, w1, b1, b2, b3, packed, max_mean, feat, argmax, h2_save=None):
    _req(x, w1, b1, b2, b3, packed, feat, argmax)
    B = x.shape[0]
    with TIMER.bracket("pointnet_enc_fwd"):
        check(lib.pm_pointnet_enc_fwd_bf6(_ptr(x), _rows(x, "x"), B, P, Cc, int(sub_mean), _ptr(w1), _ptr(b1), _ptr(b2),
                                          _ptr(b3), _ptr(packed), int(max_mean), _ptr(feat), _rows(feat, "feat"),
                                          _ptr(argmax), _ptr(h2_save), _stream()), "pm_pointnet_enc_fwd_bf6")


def pointnet_enc_bwd(x, P, Cc, sub_mean, w1, b1, b2, w3, packed, max_mean, dfeat, argmax, dw1, db1, dw2, db2, dw3, db3,
                     ws, h2_saved=None, act=None):
    _req(x, w1, b1, b2, w3, packed, dfeat, argmax, dw1, db1, dw2, db2, dw3, db3, h2_saved)
    B = x.shape[0]
    w = ws.get(lib.pm_pointnet_enc_bwd_workspace_bytes(B, P, Cc) + 256)
    base = w.data_ptr()
    al = (-base) % 256
    with TIMER.bracket("pointnet_enc_bwd"):
        check(lib.pm_pointnet_enc_bwd_f32(_ptr(x), _rows(x, "x"), B, P, Cc, int(sub_mean), _ptr(w1), _ptr(b1), _ptr(b2),
                                          _ptr(w3), _ptr(packed), int(max_mean), _ptr(dfeat), _rows(dfeat, "dfeat"),
                                          _ptr(argmax), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), _ptr(dw3),
                                          _ptr(db3), _ptr(h2_saved), int(ACT_TANH if act is None else act), base + al,
                                          w.numel() - al, _stream()),
              "pm_pointnet_enc_bwd_f32")


def pointnet_pack_bwd_bf6(w2, packed):
    _req(w2, packed)
    check(lib.pm_pointnet_pack_weights_bwd_bf6(_ptr(w2), _ptr(packed), _stream()), "pm_pointnet_pack_weights_bwd_bf6")


def pointnet_enc_bwd_bf6(x, P, Cc, sub_mean, w1, b1, b2, w3, packed, packed_w2, max_mean, dfeat, argmax, dw1, db1, dw2, db2, dw3,
                         db3, ws, h2_saved):
    """pointnet_enc_bwd with dW2 / dh1 on split-bf16 MFMAs (three planes, six products); tanh, saved layer 2."""
    _req(x, w1, b1, b2, w3, packed, packed_w2, dfeat, argmax, dw1, db1, dw2, db2, dw3, db3, h2_saved)
    B = x.shape[0]
    w = ws.get(lib.pm_pointnet_enc_bwd_workspace_bytes(B, P, Cc) + 256)
    base = w.data_ptr()
    al = (-base) % 256
    with TIMER.bracket("pointnet_enc_bwd"):
        check(lib.pm_pointnet_enc_bwd_bf6(_ptr(x), _rows(x, "x"), B, P, Cc, int(sub_mean), _ptr(w1), _ptr(b1), _ptr(b2),
                                          _ptr(w3), _ptr(packed), _ptr(packed_w2), int(max_mean), _ptr(dfeat),
                                          _rows(dfeat, "dfeat"), _ptr(argmax), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2),
                                          _ptr(dw3), _ptr(db3), _ptr(h2_saved), base + al, w.numel() - al, _stream()),
              "pm_pointnet_enc_bwd_bf6")


# ----------------------------------------------------------------------------- K8-K11
def ppo_actor_loss(mu, log_std, actions, old_logp, adv, old_mu, old_sigma, max_action, act_tanh, eps_clip, desired_kl,
                   adv_moments, adv_count, scal, dmu, dlog_std, ws):
    _req(mu, log_std, actions, old_logp, adv, old_mu, old_sigma, scal, dmu, dlog_std)
    B, A = mu.shape
    w = ws.get(lib.pm_ppo_actor_loss_workspace_bytes(B))
    check(lib.pm_ppo_actor_loss_fwd_bwd_f32(_ptr(mu), _rows(mu, "mu"), _ptr(log_std), _ptr(actions),
                                            _rows(actions, "actions"), _ptr(old_logp), _ptr(adv), _ptr(old_mu),
                                            _rows(old_mu, "old_mu"), _ptr(old_sigma), _rows(old_sigma, "old_sigma"),
                                            B, A, float(max_action), int(act_tanh), float(eps_clip), float(desired_kl),
                                            _ptr(adv_moments), float(adv_count), _ptr(scal), _ptr(dmu),
                                            _rows(dmu, "dmu"), _ptr(dlog_std), _ptr(w), w.numel(), _stream()),
          "pm_ppo_actor_loss_fwd_bwd_f32")


def ppo_actor_head_supported(h, w, dh):
    return bool(lib.pm_ppo_actor_head_supported(_ptr(h), _rows(h, "h"), _ptr(w), _rows(w, "w"), w.shape[0], w.shape[1], _ptr(dh),
                                                _rows(dh, "dh")))


def ppo_actor_head(h, w, b, hidden_act, log_std, actions, old_logp, adv, old_mu, old_sigma, max_action, act_tanh, eps_clip,
                   desired_kl, adv_moments, adv_count, scal, dmu, dh, dlog_std, ws, mu_out=None):
    """Policy head forward + PPO actor loss + dmu + head data gradient in one launch (pm_ppo_actor_head_f32); `ws` also keeps
    the launch's self-resetting work-group counter."""
    _req(h, w, b, log_std, actions, old_logp, adv, old_mu, old_sigma, scal, dmu, dh, dlog_std, mu_out)
    B, K = h.shape
    A = w.shape[0]
    buf = ws.get(lib.pm_ppo_actor_loss_workspace_bytes(B))
    ctr = getattr(ws, "counter", None)
    if ctr is None:
        ctr = ws.counter = torch.zeros(4, dtype=torch.int32, device=h.device)
    check(lib.pm_ppo_actor_head_f32(_ptr(h), _rows(h, "h"), _ptr(w), _rows(w, "w"), _ptr(b), K, int(hidden_act), _ptr(log_std),
                                    _ptr(actions), _rows(actions, "actions"), _ptr(old_logp), _ptr(adv), _ptr(old_mu),
                                    _rows(old_mu, "old_mu"), _ptr(old_sigma), _rows(old_sigma, "old_sigma"), B, A,
                                    float(max_action), int(act_tanh), float(eps_clip), float(desired_kl), _ptr(adv_moments),
                                    float(adv_count), _ptr(scal), _ptr(mu_out), _rows(mu_out, "mu_out") if mu_out is not None else 0,
                                    _ptr(dmu), _rows(dmu, "dmu"), _ptr(dh), _rows(dh, "dh"), _ptr(dlog_std), _ptr(buf), buf.numel(),
                                    _ptr(ctr), _stream()), "pm_ppo_actor_head_f32")


def gaussian_logp(mu, log_std, actions, max_action, act_tanh, logp, entropy):
    _req(mu, log_std, actions, logp, entropy)
    B, A = mu.shape
    check(lib.pm_gaussian_logp_f32(_ptr(mu), _rows(mu, "mu"), _ptr(log_std), _ptr(actions), _rows(actions, "actions"),
                                   B, A, float(max_action), int(act_tanh), _ptr(logp), _ptr(entropy), _stream()),
          "pm_gaussian_logp_f32")


def gaussian_logp_bwd(mu, log_std, actions, max_action, act_tanh, dlogp, dent, dmu, dlog_std):
    _req(mu, log_std, actions, dlogp, dent, dmu, dlog_std)
    B, A = mu.shape
    check(lib.pm_gaussian_logp_bwd_f32(_ptr(mu), _rows(mu, "mu"), _ptr(log_std), _ptr(actions), _rows(actions, "actions"), B, A,
                                       float(max_action), int(act_tanh), _ptr(dlogp), _ptr(dent), _ptr(dmu),
                                       _rows(dmu, "dmu") if dmu is not None else 0, _ptr(dlog_std), _stream()),
          "pm_gaussian_logp_bwd_f32")


def action_activation_bwd(out, dout, dmu, max_action, act_tanh):
    _req(out, dout, dmu)
    for t_, n_ in ((out, "out"), (dout, "dout"), (dmu, "dmu")):
        _f32c(t_, n_)
    check(lib.pm_action_activation_bwd_f32(_ptr(out), _ptr(dout), _ptr(dmu), out.numel(), float(max_action), int(act_tanh),
                                           _stream()), "pm_action_activation_bwd_f32")


def value_loss(v, returns, old_values, clipped, eps_clip, clip_mean_extern, grad_scale, scal, dv):
    _req(v, returns, old_values, clip_mean_extern, scal, dv)
    check(lib.pm_value_loss_fwd_bwd_f32(_ptr(v), _ptr(returns), _ptr(old_values), v.numel(), int(clipped),
                                        float(eps_clip), _ptr(clip_mean_extern), float(grad_scale), _ptr(scal), _ptr(dv),
                                        dv.stride(0) if dv.dim() == 2 else 1, _stream()), "pm_value_loss_fwd_bwd_f32")


def value_head_supported(h, w, dh):
    return w.shape[0] == 1 and w.is_contiguous() and bool(lib.pm_value_head_supported(_ptr(h), _rows(h, "h"), _ptr(w), w.shape[1], _ptr(dh),
                                                                                      _rows(dh, "dh")))


def value_head(h, w, b, hidden_act, returns, old_values, clipped, eps_clip, clip_mean_extern, grad_scale, scal, dv, dh, ws, v_out=None):
    """Value head forward + value loss + dV + head data gradient in one launch (pm_value_head_f32)."""
    _req(h, w, b, returns, old_values, clip_mean_extern, scal, dv, dh, v_out)
    B, K = h.shape
    buf = ws.get(lib.pm_value_head_workspace_bytes())
    ctr = getattr(ws, "counter", None)
    if ctr is None:
        ctr = ws.counter = torch.zeros(4, dtype=torch.int32, device=h.device)
    check(lib.pm_value_head_f32(_ptr(h), _rows(h, "h"), _ptr(w), _ptr(b), K, int(hidden_act), _ptr(returns), _ptr(old_values), B,
                                int(clipped), float(eps_clip), _ptr(clip_mean_extern), float(grad_scale), _ptr(scal), _ptr(v_out),
                                _ptr(dv), dv.stride(0) if dv.dim() == 2 else 1, _ptr(dh), _rows(dh, "dh"), _ptr(buf), buf.numel(),
                                _ptr(ctr), _stream()), "pm_value_head_f32")


def mse_tanh_loss(stu_mu, tea_mu, max_action, act_tanh, grad_scale, scal, dstu):
    _req(stu_mu, tea_mu, scal, dstu)
    B, A = stu_mu.shape
    check(lib.pm_mse_tanh_loss_fwd_bwd_f32(_ptr(stu_mu), _rows(stu_mu, "stu_mu"), _ptr(tea_mu), _rows(tea_mu, "tea_mu"),
                                           B, A, float(max_action), int(act_tanh), float(grad_scale), _ptr(scal),
                                           _ptr(dstu), _rows(dstu, "dstu"), _stream()), "pm_mse_tanh_loss_fwd_bwd_f32")


def action_activation(mu, out, max_action, act_tanh):
    _req(mu, out)
    _f32c(mu, "mu")
    _f32c(out, "out")
    check(lib.pm_action_activation_f32(_ptr(mu), _ptr(out), mu.numel(), float(max_action), int(act_tanh), _stream()),
          "pm_action_activation_f32")


def clip_adam_step(p, g, m, v, n_clip, max_norm, lr, b1, b2, eps, state, skip_flag, gnorm_out, ws):
    _req(p, g, m, v, state, skip_flag, gnorm_out)
    n = p.numel()
    w = ws.get(lib.pm_clip_adam_workspace_bytes(n))
    check(lib.pm_clip_adam_step_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, int(n_clip), float(max_norm), float(lr),
                                    float(b1), float(b2), float(eps), _ptr(state), _ptr(skip_flag), _ptr(gnorm_out),
                                    _ptr(w), w.numel(), _stream()), "pm_clip_adam_step_f32")


def ppo_accumulate_stats(acc, scal, which):
    _req(acc, scal)
    check(lib.pm_ppo_accumulate_stats_f32(_ptr(acc), _ptr(scal), int(which), _stream()), "pm_ppo_accumulate_stats_f32")


# ----------------------------------------------------------------------------- K12-K14
def fps(xyz, K, ws):
    """xyz (B,P,D) -> idx (B,K) int32 (pytorch3d sample_farthest_points defaults)."""
    _req(xyz)
    _f32c(xyz, "xyz")
    B, P, Dd = xyz.shape
    idx = torch.empty(B, K, dtype=torch.int32, device=xyz.device)
    nb = lib.pm_fps_workspace_bytes(B, P)
    w = ws.get(nb) if nb else None
    check(lib.pm_fps_f32(_ptr(xyz), B, P, Dd, K, _ptr(idx), _ptr(w), w.numel() if w is not None else 0, _stream()),
          "pm_fps_f32")
    return idx


def ball_query(xyz, centers, radius, nsample):
    _req(xyz, centers)
    _f32c(xyz, "xyz")
    _f32c(centers, "centers")
    B, P, _ = xyz.shape
    S = centers.shape[1]
    idx = torch.empty(B, S, nsample, dtype=torch.int32, device=xyz.device)
    check(lib.pm_ball_query_f32(_ptr(xyz), _ptr(centers), B, P, S, float(radius), nsample, _ptr(idx), _stream()),
          "pm_ball_query_f32")
    return idx


def group_points(feat, idx):
    _req(feat, idx)
    _f32c(feat, "feat")
    B, P, Cc = feat.shape
    S, ns = idx.shape[1], idx.shape[2]
    out = torch.empty(B, S, ns, Cc, dtype=torch.float32, device=feat.device)
    check(lib.pm_group_points_f32(_ptr(feat), _ptr(idx), B, P, Cc, S, ns, _ptr(out), _stream()), "pm_group_points_f32")
    return out


def group_points_bwd(dout, idx, P):
    _req(dout, idx)
    _f32c(dout, "dout")
    B, S, ns, Cc = dout.shape
    dfeat = torch.zeros(B, P, Cc, dtype=torch.float32, device=dout.device)
    check(lib.pm_group_points_bwd_f32(_ptr(dout), _ptr(idx), B, P, Cc, S, ns, _ptr(dfeat), _stream()),
          "pm_group_points_bwd_f32")
    return dfeat


# ----------------------------------------------------------------------------- K15 (PointNet++ glue)
def group_concat(xyz, feat, centers, idx, ldo):
    """-> (B*S*ns, ldo) rows [xyz[idx]-center | feat[idx] | 0-pad]."""
    _req(xyz, feat, centers, idx)
    B, P, _ = xyz.shape
    S, ns = idx.shape[1], idx.shape[2]
    Cf = 0 if feat is None else feat.shape[2]
    out = torch.empty(B * S * ns, ldo, dtype=torch.float32, device=xyz.device)
    check(lib.pm_group_concat_f32(_ptr(xyz), _ptr(feat), _ptr(centers), _ptr(idx), B, P, Cf, S, ns, ldo, _ptr(out),
                                  _stream()), "pm_group_concat_f32")
    return out


def group_concat_bwd(dout, idx, B, P, Cf, ldo):
    _req(dout, idx)
    S, ns = idx.shape[1], idx.shape[2]
    dfeat = torch.zeros(B, P, Cf, dtype=torch.float32, device=dout.device)
    check(lib.pm_group_concat_bwd_f32(_ptr(dout), _ptr(idx), B, P, Cf, S, ns, ldo, _ptr(dfeat), _stream()),
          "pm_group_concat_bwd_f32")
    return dfeat


def maxpool_rows(x, G, ns, out):
    """x (G*ns, C) -> out (G, C) view (any row stride), arg (G, C) int32."""
    _req(x, out)
    _f32c(x, "x")
    Cc = x.shape[1]
    arg = torch.empty(G, Cc, dtype=torch.int32, device=x.device)
    check(lib.pm_maxpool_rows_f32(_ptr(x), G, ns, Cc, _ptr(out), _rows(out, "out"), _ptr(arg), _stream()),
          "pm_maxpool_rows_f32")
    return arg


def maxpool_rows_bwd(dout, arg, ns, y_tanh=None):
    """dx (G*ns, C); `y_tanh` = the pooled (G*ns, C) tanh outputs -> the activation derivative is fused in."""
    _req(dout, arg, y_tanh)
    G, Cc = arg.shape
    dx = torch.empty(G * ns, Cc, dtype=torch.float32, device=dout.device)
    check(lib.pm_maxpool_rows_bwd_f32(_ptr(dout), _rows(dout, "dout"), _ptr(arg), G, ns, Cc, _ptr(y_tanh), _ptr(dx),
                                      _stream()), "pm_maxpool_rows_bwd_f32")
    return dx


# ----------------------------------------------------------------------------- SURVEY 8(b) names: one-call forms
def adv_normalize(x, ws, eps=1e-8):
    """In place x <- (x - mean) / (std_unbiased + eps) over all elements (storage.py:114, single process): pm_adv_normalize_f32."""
    _req(x)
    _f32c(x, "x")
    n = x.numel()
    w = ws.get(int(lib.pm_adv_normalize_workspace_bytes(n)) + 8)
    base = w.data_ptr()
    al = (-base) % 8
    check(lib.pm_adv_normalize_f32(_ptr(x), n, float(eps), base + al, w.numel() - al, _stream()), "pm_adv_normalize_f32")
    return x


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[0 if t is None else t.data_ptr() for t in ts])


def mlp_fwd(x, weights, biases, act):
    """network.py:27-54 `MLP` forward through pm_mlp_fwd_f32 -> list of the layer outputs (the last one is the network output)."""
    _req(x, *weights, *biases)
    M = x.shape[0]
    dims = [weights[0].shape[1]] + [w.shape[0] for w in weights]
    hs = [torch.empty(M, d, dtype=torch.float32, device=x.device) for d in dims[1:]]
    check(lib.pm_mlp_fwd_f32(_ptr(x), _rows(x, "x"), M, len(weights), (C.c_int * len(dims))(*dims), _ptr_array(weights),
                             _ptr_array(biases), act, _ptr_array(hs), _stream()), "pm_mlp_fwd_f32")
    return hs


def mlp_bwd(x, weights, hs, act, dy, ws, need_dx=False):
    """-> (dW list, db list, dX or None) through pm_mlp_bwd_f32."""
    _req(x, dy, *weights, *hs)
    M = x.shape[0]
    dims = [weights[0].shape[1]] + [w.shape[0] for w in weights]
    cd = (C.c_int * len(dims))(*dims)
    dws = [torch.empty_like(w) for w in weights]
    dbs = [torch.empty(w.shape[0], dtype=torch.float32, device=x.device) for w in weights]
    dx = torch.empty(M, dims[0], dtype=torch.float32, device=x.device) if need_dx else None
    w = ws.get(int(lib.pm_mlp_bwd_workspace_bytes(M, len(weights), cd)) + 16)
    base = w.data_ptr()
    al = (-base) % 16
    check(lib.pm_mlp_bwd_f32(_ptr(x), _rows(x, "x"), M, len(weights), cd, _ptr_array(weights), _ptr_array(hs), act, _ptr(dy),
                             _ptr_array(dws), _ptr_array(dbs), _ptr(dx), base + al, w.numel() - al, _stream()), "pm_mlp_bwd_f32")
    return dws, dbs, dx


# ----------------------------------------------------------------------------- K15 fused SA level
def sa_supported(C1, C2, C3, ns):
    return bool(lib.pm_sa_supported(C1, C2, C3, ns))


def sa_pack(w2, w3, packed):
    _req(w2, w3, packed)
    C2, C1 = w2.shape
    C3 = w3.shape[0]
    _f32c(w2, "w2")
    _f32c(w3, "w3")
    check(lib.pm_sa_pack_weights_f32(_ptr(w2), _ptr(w3), C1, C2, C3, _ptr(packed), _stream()),
          "pm_sa_pack_weights_f32")


def sa_fwd(xyz, centers, idx, Y, w1, b1, b2, b3, packed, dims, pooled, h2_save=None):
    """xyz (B,P,3), centers (B,S,3), idx (B,S,32) int32, Y (B*P,C1) or None -> pooled (B*S, C3) view, arg int32."""
    _req(xyz, centers, idx, Y, w1, packed, pooled)
    B, P, _ = xyz.shape
    S, ns = idx.shape[1], idx.shape[2]
    C1, C2, C3 = dims
    _f32c(xyz, "xyz")
    _f32c(centers, "centers")
    arg = torch.empty(B * S, C3, dtype=torch.int32, device=xyz.device)
    with TIMER.bracket(f"sa_fwd_{C1}x{C2}x{C3}"):
        check(lib.pm_sa_fwd_f32(_ptr(xyz), _ptr(centers), _ptr(idx), _ptr(Y), B, P, S, ns,
                                _ptr(w1), _rows(w1, "w1"), _ptr(b1), _ptr(b2), _ptr(b3), _ptr(packed), C1, C2, C3,
                                _ptr(pooled), _rows(pooled, "pooled"), _ptr(arg), _ptr(h2_save), _stream()), "pm_sa_fwd_f32")
    return arg


def sa_bwd(xyz, centers, idx, Y, w1, b1, b2, w3, packed, dims, pooled, arg, dpooled, dw1, db1, dw2, db2, dw3, db3, dY, ws,
           h2_saved=None):
    _req(xyz, centers, idx, Y, w1, w3, packed, pooled, arg, dpooled, dw1, dw2, dw3, dY)
    B, P, _ = xyz.shape
    S, ns = idx.shape[1], idx.shape[2]
    C1, C2, C3 = dims
    _f32c(w3, "w3")
    w = ws.get(lib.pm_sa_bwd_workspace_bytes(C1, C2, C3))
    with TIMER.bracket(f"sa_bwd_{C1}x{C2}x{C3}"):
        check(lib.pm_sa_bwd_f32(_ptr(xyz), _ptr(centers), _ptr(idx), _ptr(Y), B, P, S, ns, _ptr(w1), _rows(w1, "w1"),
                                _ptr(b1), _ptr(b2), _ptr(w3), _ptr(packed), C1, C2, C3, _ptr(pooled),
                                _rows(pooled, "pooled"), _ptr(arg), _ptr(dpooled), _rows(dpooled, "dpooled"), _ptr(dw1),
                                _rows(dw1, "dw1"), _ptr(db1), _ptr(dw2), _ptr(db2), _ptr(dw3), _ptr(db3), _ptr(dY),
                                _ptr(h2_saved), _ptr(w), w.numel(), _stream()), "pm_sa_bwd_f32")


# ---- group-all level: last layer + max over the cloud in one kernel, structured backward (csrc/sa_groupall.hip)
def sa_groupall_supported(ck, co, rows):
    return bool(lib.pm_sa_groupall_supported(int(ck), int(co), int(rows)))


def sa_groupall_pack(w, packed):
    _req(w, packed)
    _f32c(w, "w")
    check(lib.pm_sa_groupall_pack_f32(_ptr(w), w.shape[1], w.shape[0], _ptr(packed), _stream()), "pm_sa_groupall_pack_f32")


def sa_groupall_fwd(h, B, R, bias, packed, feat):
    """h (B*R, CK) -> feat (B, CO) view (any row stride) = max over the cloud's rows of tanh(h W^T + bias); arg (B, CO) int32."""
    _req(h, bias, packed, feat)
    _f32c(h, "h")
    ck, co = h.shape[1], feat.shape[1]
    arg = torch.empty(B, co, dtype=torch.int32, device=h.device)
    with TIMER.bracket("sa_groupall_fwd"):
        check(lib.pm_sa_groupall_fwd_f32(_ptr(h), B, R, ck, co, _ptr(bias), _ptr(packed), _ptr(feat), _rows(feat, "feat"), _ptr(arg),
                                         _stream()), "pm_sa_groupall_fwd_f32")
    return arg


def sa_groupall_bwd(dfeat, feat, arg, w, h, B, R, dh, dw, db, ws):
    """dh (B*R, CK) = d loss / d (pre-activation of h), every row written; dw (CO, CK), db (CO) overwritten."""
    _req(dfeat, feat, arg, w, h, dh, dw, db)
    _f32c(w, "w")
    _f32c(h, "h")
    _f32c(dh, "dh")
    _f32c(dw, "dw")
    co, ck = w.shape
    wsb = ws.get(int(lib.pm_sa_groupall_bwd_workspace_bytes(B, ck, co)) + 16)
    base = wsb.data_ptr()
    al = (-base) % 16
    with TIMER.bracket("sa_groupall_bwd"):
        check(lib.pm_sa_groupall_bwd_f32(_ptr(dfeat), _rows(dfeat, "dfeat"), _ptr(feat), _rows(feat, "feat"), _ptr(arg), _ptr(w), _ptr(h),
                                         B, R, ck, co, _ptr(dh), _ptr(dw), _ptr(db), base + al, wsb.numel() - al, _stream()),
              "pm_sa_groupall_bwd_f32")


# ---- duplicate-free ("packed") form: the level over each group's DISTINCT rows (ball query pads with copies of the first hit)
class SaPlan:
    """Packed-row tables of one neighbourhood table (pm_sa_plan_i32): nothing here is read by the host."""
    __slots__ = ("grow", "rowmap", "relxyz", "tiles", "totals", "B", "P", "S", "ns", "dims", "ready", "inv_start", "inv_rows")

    def counts(self):
        """(packed rows, tiles) -- a host read; diagnostics / buffer sizing only."""
        t = self.totals.cpu()
        return int(t[0]), int(t[1])

    def trim(self, counts=None):
        """Drop the worst-case capacity of the tables: for plans that are kept, e.g. per mini-batch slice.  counts: the (packed rows,
        tiles) a caller has already read (several plans, one host read); None: read here."""
        R, T = counts if counts is not None else self.counts()
        self.rowmap = self.rowmap[:max(R, 1)].clone()
        self.relxyz = self.relxyz[:max(R, 1)].clone()
        self.tiles = self.tiles[:max(T, 1)].clone()
        if self.inv_rows is not None:
            self.inv_rows = self.inv_rows[:max(R, 1)].clone()
        return self


def sa_packed_tile(dims):
    r, g = C.c_int(0), C.c_int(0)
    check(lib.pm_sa_packed_tile(dims[0], dims[1], dims[2], C.byref(r), C.byref(g)), "pm_sa_packed_tile")
    return r.value, g.value


def sa_plan(idx, xyz, centers, dims, ws, inverse=False):
    """idx (B, S, 32) int32 ball-query table of centres (B, S, 3) over xyz (B, P, 3) -> SaPlan for the fused level `dims`.
    inverse: also list every source point's packed rows in ascending order (pm_sa_plan_inverse_i32) -- what the deterministic
    gradient of a level with input features sums over (sa_dy_segsum)."""
    _req(idx, xyz, centers)
    _f32c(xyz, "xyz")
    _f32c(centers, "centers")
    B, S, ns = idx.shape
    P = xyz.shape[1]
    if idx.dtype != torch.int32 or not idx.is_contiguous():
        raise TypeError("sa_plan: idx must be a contiguous int32 tensor")
    tr, tg = sa_packed_tile(dims)
    G = B * S
    pl = SaPlan()
    pl.ready = None                       # set by a caller that shares the plan between streams (an event recorded after the build)
    pl.B, pl.P, pl.S, pl.ns, pl.dims = B, P, S, ns, tuple(dims)
    dev = idx.device
    pl.grow = torch.empty(G + 1, dtype=torch.int32, device=dev)
    pl.rowmap = torch.empty(G * ns, 2, dtype=torch.int32, device=dev)
    pl.relxyz = torch.empty(G * ns, 4, dtype=torch.float32, device=dev)
    pl.tiles = torch.empty(G, 4, dtype=torch.int32, device=dev)
    pl.totals = torch.zeros(4, dtype=torch.int32, device=dev)
    nb = int(lib.pm_sa_plan_workspace_bytes(B, S))
    w = ws.get(nb + 16)
    base = w.data_ptr()
    al = (-base) % 16
    with TIMER.bracket("sa_plan"):
        check(lib.pm_sa_plan_i32(_ptr(idx), _ptr(xyz), _ptr(centers), B, P, S, ns, tr, tg, _ptr(pl.grow), _ptr(pl.rowmap),
                                 _ptr(pl.relxyz), _ptr(pl.tiles), _ptr(pl.totals), base + al, w.numel() - al, _stream()), "pm_sa_plan_i32")
        pl.inv_start = pl.inv_rows = None
        if inverse:
            pl.inv_start = torch.empty(B * P + 1, dtype=torch.int32, device=dev)
            pl.inv_rows = torch.empty(G * ns, dtype=torch.int32, device=dev)
            check(lib.pm_sa_plan_inverse_i32(_ptr(pl.rowmap), _ptr(pl.grow), B, P, S, _ptr(pl.inv_start), _ptr(pl.inv_rows), _stream()),
                  "pm_sa_plan_inverse_i32")
    return pl


def sa_dy_segsum(plan, dz1, dY):
    """dY (B*P, C1) = per source point the sum of its packed rows of dz1 (R, C1) in ascending row order (no atomics, no zero-fill)."""
    _req(dz1, dY)
    _f32c(dz1, "dz1")
    if plan.inv_start is None:
        raise ValueError("sa_dy_segsum: the plan was built without its inverse (sa_plan(..., inverse=True))")
    if dY.shape[0] != plan.B * plan.P or dz1.shape[1] != dY.shape[1]:
        raise ValueError("sa_dy_segsum: shapes do not match the plan")
    with TIMER.bracket("sa_dy_segsum"):
        check(lib.pm_sa_dy_segsum_f32(_ptr(dz1), _ptr(plan.inv_start), _ptr(plan.inv_rows), plan.B * plan.P, dz1.shape[1], _ptr(dY),
                                      _rows(dY, "dY"), _stream()), "pm_sa_dy_segsum_f32")
    return dY


def gather_copy(dst, src, table):
    """dst[q] = src[table[q]] (0 where table[q] < 0): all weight-derived operand copies of a network in one launch."""
    _req(dst, src, table)
    _f32c(dst, "dst")
    _f32c(src, "src")
    _i32c(table, "table")
    if table.numel() != dst.numel():
        raise ValueError("gather_copy: one table entry per destination element")
    check(lib.pm_gather_copy_f32(_ptr(dst), _ptr(src), _ptr(table), dst.numel(), _stream()), "pm_gather_copy_f32")
    return dst


def sa_fwd_packed(plan, Y, w1, b1, b2, b3, packed, dims, pooled, h2_save=None, tail_xyz=None):
    """tail_xyz (B*S, 3): the columns of `pooled`'s rows behind the C3 features become (x, y, z, 0 ...) -- pooled is then a view of
    a group-all level's input rows and needs no tail copy."""
    _req(Y, w1, packed, pooled, tail_xyz)
    if tail_xyz is not None:
        _f32c(tail_xyz, "tail_xyz")
        if tuple(tail_xyz.shape) != (plan.B * plan.S, 3):
            raise ValueError("sa_fwd_packed: tail_xyz must be (B*S, 3)")
    B, P, S = plan.B, plan.P, plan.S
    C1, C2, C3 = dims
    if tuple(dims) != plan.dims or pooled.shape[0] != B * S or (Y is not None and Y.shape[0] != B * P):
        raise ValueError("sa_fwd_packed: the plan was built for another batch / level shape")
    arg = torch.empty(B * S, C3, dtype=torch.int32, device=pooled.device)
    with TIMER.bracket(f"sa_fwd_{C1}x{C2}x{C3}"):
        check(lib.pm_sa_fwd_packed_f32(_ptr(Y), B, P, S, _ptr(plan.grow), _ptr(plan.rowmap), _ptr(plan.relxyz),
                                       _ptr(plan.tiles), _ptr(plan.totals), _ptr(w1), _rows(w1, "w1"), _ptr(b1), _ptr(b2),
                                       _ptr(b3), _ptr(packed), C1, C2, C3, _ptr(pooled), _rows(pooled, "pooled"), _ptr(arg),
                                       _ptr(h2_save), _ptr(tail_xyz), (_rows(pooled, "pooled") - C3) if tail_xyz is not None else 0,
                                       _stream()), "pm_sa_fwd_packed_f32")
    return arg


def sa_dy_consume_supported(C1, cf):
    return bool(lib.pm_sa_dy_consume_supported(int(C1), int(cf)))


def sa_dy_consume_pack(w1, cf, packed):
    """w1 (C1, >= 3 + cf): its feature columns in MFMA operand order (after every update)."""
    _req(w1, packed)
    check(lib.pm_sa_dy_consume_pack_f32(_ptr(w1), _rows(w1, "w1"), w1.shape[0], cf, _ptr(packed), _stream()), "pm_sa_dy_consume_pack_f32")
    return packed


def sa_dy_consume(plan, dz1, feat, packed_w1f, dfeat, dw1, ws, dY=None):
    """dz1 (R, C1) per packed row -> dfeat (B*P, cf) = dY W1f, dw1[:, 3:3+cf] = dY^T feat (pad columns zeroed), with dY = the fixed-order
    per-source-point sums of dz1 formed in LDS (never written, unless the optional dY copy is asked for)."""
    _req(dz1, feat, packed_w1f, dfeat, dw1, dY)
    _f32c(dz1, "dz1")
    C1, cf = dz1.shape[1], feat.shape[1]
    npts = plan.B * plan.P
    if plan.inv_start is None or feat.shape[0] != npts or (dfeat is not None and tuple(dfeat.shape) != (npts, cf)) or dw1.shape[0] != C1:
        raise ValueError("sa_dy_consume: shapes do not match the plan (built with inverse=True?)")
    w = ws.get(lib.pm_sa_dy_consume_workspace_bytes(C1, cf))
    with TIMER.bracket("sa_dy_consume"):
        check(lib.pm_sa_dy_consume_f32(_ptr(dz1), _ptr(plan.inv_start), _ptr(plan.inv_rows), npts, C1, cf, _ptr(feat), _rows(feat, "feat"),
                                       _ptr(packed_w1f), _ptr(dfeat), _rows(dfeat, "dfeat") if dfeat is not None else 0, _ptr(dw1),
                                       _rows(dw1, "dw1"), dw1.shape[1], _ptr(dY), _rows(dY, "dY") if dY is not None else 0, _ptr(w),
                                       w.numel(), _stream()), "pm_sa_dy_consume_f32")
    return dfeat


def sa_bwd_packed(plan, Y, w1, b1, b2, w3, packed, dims, pooled, arg, dpooled, dw1, db1, dw2, db2, dw3, db3, dY, ws, h2_saved=None,
                  dz1=None, zero_pad_cols=False):
    """dz1 (R, C1): the layer-1 gradient per packed row (plain stores; sum it per source point with sa_dy_segsum) -- the
    deterministic path; dY (B*P, C1) zero-filled: the same sums by fp32 atomics (run-dependent last bits; kept for A/B).
    zero_pad_cols: a level without input features -- columns 3.. of dw1 are padding and are zeroed by the reduction launch."""
    _req(Y, w1, w3, packed, pooled, arg, dpooled, dw1, dw2, dw3, dY, dz1)
    if dz1 is not None:
        _f32c(dz1, "dz1")
        if dz1.shape[1] != dims[0]:
            raise ValueError("sa_bwd_packed: dz1 must be (R, C1)")
    B, P, S = plan.B, plan.P, plan.S
    C1, C2, C3 = dims
    if tuple(dims) != plan.dims or pooled.shape[0] != B * S:
        raise ValueError("sa_bwd_packed: the plan was built for another batch / level shape")
    _f32c(w3, "w3")
    w = ws.get(lib.pm_sa_bwd_workspace_bytes(C1, C2, C3))
    with TIMER.bracket(f"sa_bwd_{C1}x{C2}x{C3}"):
        check(lib.pm_sa_bwd_packed_f32(_ptr(Y), B, P, S, _ptr(plan.grow), _ptr(plan.rowmap), _ptr(plan.relxyz),
                                       _ptr(plan.tiles), _ptr(plan.totals), _ptr(w1), _rows(w1, "w1"), _ptr(b1), _ptr(b2),
                                       _ptr(w3), _ptr(packed), C1, C2, C3, _ptr(pooled), _rows(pooled, "pooled"), _ptr(arg),
                                       _ptr(dpooled), _rows(dpooled, "dpooled"), _ptr(dw1), _rows(dw1, "dw1"), _ptr(db1),
                                       _ptr(dw2), _ptr(db2), _ptr(dw3), _ptr(db3), _ptr(dY), _ptr(dz1), dw1.shape[1] if zero_pad_cols else 0,
                                       _ptr(h2_saved), _ptr(w), w.numel(), _stream()), "pm_sa_bwd_packed_f32")


# ----------------------------------------------------------------------------- sparse-voxel U-Net blocks
def voxel_grid0(x, P, C, R):
    """x (B, >= P*C) rows of P points (x, y, z, f...) -> (grid (B*R^3) i32, coords (B*P, 4) i32, feat (B*P, 4) f32)."""
    _req(x)
    B = x.shape[0]
    grid = torch.empty(B * R ** 3, dtype=torch.int32, device=x.device)
    coords = torch.empty(B * P, 4, dtype=torch.int32, device=x.device)
    feat = torch.empty(B * P, 4, dtype=torch.float32, device=x.device)
    check(lib.pm_voxel_grid0_f32(_ptr(x), _rows(x, "x"), B, P, C, R, _ptr(grid), _ptr(coords), _ptr(feat), _stream()),
          "pm_voxel_grid0_f32")
    return grid, coords, feat


def voxel_nbr27(coords, grid, R, taps=27):
    """(rows, 27) neighbour table; taps > 27: the rows are `taps` wide and the extra columns hold -1 (absent taps), returned
    as the (rows, taps) table -- its [:, :27] view is the plain table."""
    _req(coords, grid)
    rows = coords.shape[0]
    nbr = torch.empty(rows, taps, dtype=torch.int32, device=coords.device)
    check(lib.pm_voxel_nbr27_i32(_ptr(coords), rows, _ptr(grid), R, _ptr(nbr), int(taps), _stream()), "pm_voxel_nbr27_i32")
    return nbr


def voxel_mirror27(nbr):
    """The mirrored 27-neighbour table of the data gradient (pm_voxel_mirror27_i32)."""
    _req(nbr)
    out = torch.empty_like(nbr)
    check(lib.pm_voxel_mirror27_i32(_ptr(nbr), nbr.shape[0], _ptr(out), _stream()), "pm_voxel_mirror27_i32")
    return out


def voxel_down(coords_f, grid_f, Rf, B):
    """The 2x strided level of a fine level: dict(R, rows, grid, coords, child (rows, 8), parent / parent_canon / slot (rows_f)).
    One host read (the level's row count sizes its buffers and GEMMs)."""
    _req(coords_f, grid_f)
    dev = coords_f.device
    Rc = (Rf + 1) // 2
    rows_f = coords_f.shape[0]
    grid_c = torch.empty(B * Rc ** 3, dtype=torch.int32, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    check(lib.pm_voxel_down_count_i32(_ptr(coords_f), rows_f, B, Rc, _ptr(grid_c), _ptr(counts), _stream()),
          "pm_voxel_down_count_i32")
    base = torch.empty(B + 1, dtype=torch.int32, device=dev)            # base[B] = the level's row count
    check(lib.pm_exclusive_scan_i32(_ptr(counts), B, _ptr(base), base.data_ptr() + 4 * B, _stream()), "pm_exclusive_scan_i32")
    rows_c = int(base[B].item())
    coords_c = torch.empty(rows_c, 4, dtype=torch.int32, device=dev)
    child = torch.empty(rows_c, 8, dtype=torch.int32, device=dev)
    parent = torch.empty(rows_f, dtype=torch.int32, device=dev)
    parent_canon = torch.empty(rows_f, dtype=torch.int32, device=dev)
    slot = torch.empty(rows_f, dtype=torch.int32, device=dev)
    check(lib.pm_voxel_down_build_i32(_ptr(coords_f), rows_f, _ptr(grid_f), Rf, B, Rc, _ptr(base), _ptr(grid_c), rows_c,
                                      _ptr(coords_c), _ptr(child), _ptr(parent), _ptr(parent_canon), _ptr(slot), _stream()),
          "pm_voxel_down_build_i32")
    return dict(R=Rc, rows=rows_c, grid=grid_c, coords=coords_c, child=child, parent=parent, parent_canon=parent_canon, slot=slot)


def rows_gather(src, idx, C, dst):
    """dst[r][j*C:(j+1)*C] = src[idx[r][j]][:C] (zeros where idx < 0); src / dst are 2-D views with unit inner stride."""
    _req(src, idx, dst)
    rows = idx.shape[0]
    J = idx.shape[1] if idx.dim() == 2 else 1
    check(lib.pm_rows_gather_f32(_ptr(src), _rows(src, "src"), _ptr(idx), rows, J, C, _ptr(dst), _rows(dst, "dst"), _stream()),
          "pm_rows_gather_f32")
    return dst


def sparse_conv_fwd(src, idx, C, w, b, y, act, zero):
    """y = act(gather(src, idx) @ w.T + b) with the gather inside the GEMM loader (J*C % 32 == 0)."""
    _req(src, idx, w, b, y, zero)
    rows, J = idx.shape
    check(lib.pm_sparse_conv_fwd_f32(_ptr(src), _rows(src, "src"), _ptr(idx), rows, J, C, _ptr(w), _rows(w, "w"), _ptr(b), _ptr(y),
                                     _rows(y, "y"), w.shape[0], int(act), _ptr(zero), _stream()), "pm_sparse_conv_fwd_f32")
    return y


def sparse_conv_bwd_data(dy, idx_t, wt, h, dx, act, zero):
    """dx = (gather(dy, idx_t) @ wt.T) * act'(h): the data gradient of a submanifold convolution through the mirrored table
    `idx_t` and the transposed weight view `wt` (C_in, J*C_out), wt[ci, j*C_out + co] = w[co, j*C_in + ci] -- see pm_sparse_conv_bwd_data_f32."""
    _req(dy, idx_t, wt, h, dx, zero)
    rows, J = idx_t.shape
    Cout = dy.shape[1]
    check(lib.pm_sparse_conv_bwd_data_f32(_ptr(dy), _rows(dy, "dy"), _ptr(idx_t), rows, J, Cout, _ptr(wt), _rows(wt, "wt"), _ptr(h),
                                          _rows(h, "h") if h is not None else 0, _ptr(dx), _rows(dx, "dx"), wt.shape[0], int(act),
                                          _ptr(zero), _stream()), "pm_sparse_conv_bwd_data_f32")
    return dx


def sparse_conv_bwd_data_scatter(dy, w, idx, C, h, dx, act):
    """dx[idx[r, j]] = (dy[r] @ w[:, j*C:(j+1)*C]) * act'(h[idx[r, j]]) for convolutions with non-overlapping patches
    (pm_sparse_conv_bwd_data_scatter_f32); w = the forward's tap-major weight (Cout, J*C)."""
    _req(dy, w, idx, h, dx)
    _f32c(dx, "dx")
    rows, J = idx.shape
    check(lib.pm_sparse_conv_bwd_data_scatter_f32(_ptr(dy), _rows(dy, "dy"), _ptr(w), _rows(w, "w"), _ptr(idx), rows, J, C,
                                                  dy.shape[1], _ptr(h), _ptr(dx), int(act), _stream()),
          "pm_sparse_conv_bwd_data_scatter_f32")
    return dx


def sparse_conv_bwd_weight(dy, src, idx, C, dw, db, zero, ws):
    _req(dy, src, idx, dw, db, zero)
    rows, J = idx.shape
    N = dy.shape[1]
    w = ws.get(lib.pm_sparse_conv_bwd_weight_workspace_bytes(rows, N, J, C) + 256)
    base = w.data_ptr()
    al = (-base) % 256
    check(lib.pm_sparse_conv_bwd_weight_f32(_ptr(dy), _rows(dy, "dy"), _ptr(src), _rows(src, "src"), _ptr(idx), rows, J, C, _ptr(dw),
                                            _rows(dw, "dw"), _ptr(db), N, _ptr(zero), base + al, w.numel() - al, _stream()),
          "pm_sparse_conv_bwd_weight_f32")


def rows_gather_bwd(dcols, tidx, C, dsrc, tslot=None, mode=0, reverse=False, self_col=-1, y_tanh=None, accumulate=False, rowmap=None,
                    skip=None):
    """rowmap (int32, one entry per row the table can name): dcols holds a subset of those rows -- rowmap[row] = its row in dcols or -1.
    skip = (values (N, C), skipmap (rows) int32): a sparse raw contribution added before the activation derivative (dsrc is overwritten)."""
    _req(dcols, tidx, dsrc, tslot, y_tanh, rowmap)
    rows = tidx.shape[0]
    J = tidx.shape[1] if tidx.dim() == 2 else 1
    if skip is not None:
        sv, sm = skip
        _req(sv, sm)
        if rowmap is not None or accumulate:
            raise ValueError("rows_gather_bwd: skip= takes neither rowmap nor accumulate")
        if sm.dtype != torch.int32 or sm.numel() < rows:
            raise TypeError("rows_gather_bwd: skipmap must be int32 with one entry per row")
        check(lib.pm_rows_gather_bwd_skip_f32(_ptr(dcols), _rows(dcols, "dcols"), _ptr(tidx), _ptr(tslot), int(mode), int(reverse),
                                              int(self_col), rows, J, C, _ptr(y_tanh), _rows(y_tanh, "y") if y_tanh is not None else 0,
                                              _ptr(dsrc), _rows(dsrc, "dsrc"), _ptr(sv), _rows(sv, "skip"), _ptr(sm), _stream()),
              "pm_rows_gather_bwd_skip_f32")
        return dsrc
    check(lib.pm_rows_gather_bwd_mapped_f32(_ptr(dcols), _rows(dcols, "dcols"), _ptr(tidx), _ptr(tslot), int(mode), int(reverse),
                                            int(self_col), rows, J, C, _ptr(y_tanh), _rows(y_tanh, "y") if y_tanh is not None else 0,
                                            int(accumulate), _ptr(dsrc), _rows(dsrc, "dsrc"), _ptr(rowmap), _stream()),
          "pm_rows_gather_bwd_mapped_f32")
    return dsrc


def col_blocks(dst, src, blocks, zero_other=True, col0=0, dst_cols=None):
    """dst[:, d : d + e - s] = src[:, s:e] for the (s, e, d) in `blocks` (at most two); the other columns of dst in [col0, dst_cols)
    are zeroed if zero_other (pm_col_blocks_f32).  dst / src: 2-D float32 views with unit inner stride; src=None with no block: zero-only.
    Blocks must land inside [col0, dst_cols) and must not overlap (the library refuses otherwise)."""
    _req(dst, src)
    if dst.dim() != 2 or dst.stride(1) != 1 or (src is not None and (src.dim() != 2 or src.stride(1) != 1 or dst.shape[0] != src.shape[0])):
        raise ValueError("col_blocks: 2-D views with unit inner stride and equal row counts")
    if src is None and blocks:
        raise ValueError("col_blocks: blocks need a source")
    b = list(blocks) + [(0, 0, 0)] * (2 - len(blocks))
    (s0, e0, d0), (s1, e1, d1) = b
    check(lib.pm_col_blocks_f32(_ptr(dst), dst.stride(0), _ptr(src), src.stride(0) if src is not None else 0, dst.shape[0],
                                dst.shape[1] if dst_cols is None else dst_cols,
                                int(col0), s0, e0, d0, s1, e1, d1, int(zero_other), _stream()), "pm_col_blocks_f32")
    return dst


def _i32c(t, name):
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise TypeError(f"{name} must be a contiguous int32 tensor")


def rows_uniq(src, S, pad_out, row_base=0, table=None, pad_in=-1):
    """src (B, >= S) int32 ids per cloud -> (u, um, rank), each (B, S) int32: the cloud's distinct ids ascending (through `table` if
    given; src == pad_in -> pad_out), padded with pad_out / -1, and every id's slot (pm_rows_uniq_i32; S <= 64)."""
    _req(src, table)
    if src.dtype != torch.int32 or src.stride(-1) != 1 or src.dim() != 2:
        raise TypeError("rows_uniq: src must be a 2-D int32 tensor with unit inner stride")
    if table is not None:
        _i32c(table, "table")
    B = src.shape[0]
    u, um, rank = (torch.empty(B, S, dtype=torch.int32, device=src.device) for _ in range(3))
    check(lib.pm_rows_uniq_i32(_ptr(src), src.stride(0), B, S, int(row_base), _ptr(table), int(pad_in), int(pad_out), _ptr(u), _ptr(um),
                               _ptr(rank), _stream()), "pm_rows_uniq_i32")
    return u, um, rank


def child_sum(x, rank, out):
    """out[b*S + j] = sum over i (ascending) with rank[b, i] == j of x[b*S + i]   (pm_child_sum_f32)."""
    _req(x, rank, out)
    _i32c(rank, "rank")
    B, S = rank.shape
    check(lib.pm_child_sum_f32(_ptr(x), _rows(x, "x"), _ptr(rank), B, S, x.shape[1], _ptr(out), _rows(out, "out"), _stream()),
          "pm_child_sum_f32")
    return out


def rowmap_scatter(n, ids, pad):
    """(n,) int32 map: -1 everywhere, map[ids[k]] = k for ids[k] != pad   (pm_rowmap_scatter_i32)."""
    _req(ids)
    _i32c(ids, "ids")
    m = torch.empty(n, dtype=torch.int32, device=ids.device)
    check(lib.pm_rowmap_scatter_i32(_ptr(m), n, _ptr(ids), ids.numel(), int(pad), _stream()), "pm_rowmap_scatter_i32")
    return m


def table_rows(table, sel):
    """Rows sel (N,) of an int32 index table (rows, J) (unit inner stride; -1 in sel -> a row of -1)   (pm_table_rows_i32)."""
    _req(table, sel)
    _i32c(sel, "sel")
    if table.dtype != torch.int32 or table.stride(1) != 1:
        raise TypeError("table_rows: table must be int32 with unit inner stride")
    out = torch.empty(sel.numel(), table.shape[1], dtype=torch.int32, device=table.device)
    check(lib.pm_table_rows_i32(_ptr(table), table.stride(0), table.shape[1], _ptr(sel), sel.numel(), _ptr(out), _stream()),
          "pm_table_rows_i32")
    return out


def voxel_vcat_table(parent, m, rows_hi):
    """(rows, m + 1) int32 gather table of a virtual [unpool | skip] operand   (pm_voxel_vcat_table_i32)."""
    _req(parent)
    _i32c(parent, "parent")
    rows = parent.numel()
    out = torch.empty(rows, m + 1, dtype=torch.int32, device=parent.device)
    check(lib.pm_voxel_vcat_table_i32(_ptr(parent), rows, int(m), int(rows_hi), _ptr(out), _stream()), "pm_voxel_vcat_table_i32")
    return out


# ----------------------------------------------------------------------------- depth -> cloud
def depth_backproject(depth, cam_pose, fx, fy, cx, cy, lo, hi):
    """depth (B,M,H,W), cam_pose (M,4,4) device -> world cloud (B, M*H*W, 3), out-of-box points zeroed."""
    import ctypes
    _req(depth, cam_pose)
    _f32c(depth, "depth")
    _f32c(cam_pose, "cam_pose")
    B, M, H, W = depth.shape
    out = torch.empty(B, M * H * W, 3, dtype=torch.float32, device=depth.device)
    lo3 = (ctypes.c_float * 3)(*[float(v) for v in lo])
    hi3 = (ctypes.c_float * 3)(*[float(v) for v in hi])
    check(lib.pm_depth_backproject_f32(_ptr(depth), B, M, H, W, _ptr(cam_pose), float(fx), float(fy), float(cx),
                                       float(cy), ctypes.cast(lo3, ctypes.c_void_p), ctypes.cast(hi3, ctypes.c_void_p),
                                       _ptr(out), _stream()), "pm_depth_backproject_f32")
    return out


# ----------------------------------------------------------------------------- Conv3D students
def conv3d_out(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


def im2col3d(x5, k, stride, pad, ldc):
    """x5: a 5-D VIEW (B, C, D, H, W) with arbitrary strides -> cols (B*Do*Ho*Wo, ldc)."""
    _req(x5)
    B, Cc, D, H, W = x5.shape
    Do, Ho, Wo = (conv3d_out(n, k, stride, pad) for n in (D, H, W))
    cols = torch.empty(B * Do * Ho * Wo, ldc, dtype=torch.float32, device=x5.device)
    check(lib.pm_im2col3d_f32(_ptr(x5), B, Cc, D, H, W, k, stride, pad, *x5.stride(), _ptr(cols), ldc, _stream()),
          "pm_im2col3d_f32")
    return cols


def conv3d_c1_supported(k, cout):
    return bool(lib.pm_conv3d_c1_supported(int(k), int(cout)))


def conv3d_c1_fwd(x5, k, stride, pad, wt, bias, act, rows=None):
    """Direct conv of a single-channel 5-D VIEW (B, 1, D, H, W) with wt (k^3, Cout) -> y (B*Do*Ho*Wo, Cout) = act(conv + b).
    rows (int64, optional): the batch is rows[b] of x5's first dimension (no gathered copy)."""
    _req(x5, wt, bias, rows)
    B, Cc, D, H, W = x5.shape
    if rows is not None:
        B = rows.numel()
    if Cc != 1:
        raise ValueError("conv3d_c1_fwd: one input channel")
    _f32c(wt, "wt")
    Do, Ho, Wo = (conv3d_out(n, k, stride, pad) for n in (D, H, W))
    cout = wt.shape[1]
    y = torch.empty(B * Do * Ho * Wo, cout, dtype=torch.float32, device=x5.device)
    sb, _, sd, sh, sw = x5.stride()
    with TIMER.bracket("conv3d_c1_fwd"):
        check(lib.pm_conv3d_c1_fwd_f32(_ptr(x5), B, D, H, W, k, stride, pad, sb, sd, sh, sw, _ptr(wt), _ptr(bias), cout, int(act),
                                       _ptr(y), cout, _ptr(rows), _stream()), "pm_conv3d_c1_fwd_f32")
    return y


def conv3d_c1_wgrad(dz, x5, k, stride, pad, dw, db, ws, rows=None):
    """dw (Cout, k^3) / db (Cout) of the direct input-layer conv from dz (rows, Cout); `rows` as in conv3d_c1_fwd."""
    _req(dz, x5, dw, db, rows)
    B, Cc, D, H, W = x5.shape
    if rows is not None:
        B = rows.numel()
    cout = dw.shape[0]
    w = ws.get(lib.pm_conv3d_c1_wgrad_workspace_bytes(cout))
    sb, _, sd, sh, sw = x5.stride()
    with TIMER.bracket("conv3d_c1_wgrad"):
        check(lib.pm_conv3d_c1_wgrad_f32(_ptr(dz), _rows(dz, "dz"), _ptr(x5), B, D, H, W, k, stride, pad, sb, sd, sh, sw, cout,
                                         _ptr(dw), _rows(dw, "dw"), _ptr(db), _ptr(rows), _ptr(w), w.numel(), _stream()),
              "pm_conv3d_c1_wgrad_f32")


def col2im3d(dcols, dx5, k, stride, pad, y_tanh5=None, act=None):
    """dcols (B*Do*Ho*Wo, ldc) -> every element of the 5-D view dx5 (B, C, D, H, W); y_tanh5 = the layer input
    (an activation output -- tanh unless `act` says otherwise -- of the same shape AND strides as dx5): its derivative is folded in."""
    _req(dcols, dx5, y_tanh5)
    B, Cc, D, H, W = dx5.shape
    if y_tanh5 is not None and (y_tanh5.shape != dx5.shape or y_tanh5.stride() != dx5.stride()):
        raise ValueError("col2im3d: y_tanh5 must be laid out like dx5")
    check(lib.pm_col2im3d_f32(_ptr(dcols), B, Cc, D, H, W, k, stride, pad, *dx5.stride(), _ptr(y_tanh5), ACT_TANH if act is None else int(act), _ptr(dx5),
                              _rows(dcols, "dcols"), _stream()), "pm_col2im3d_f32")


def tsdf_integrate(depth, pix_idx, pix_z, trunc, default_tsdf):
    """depth (B,M,H,W), pix_idx (M,V) int32, pix_z (M,V) -> (B,V)."""
    _req(depth, pix_idx, pix_z)
    _f32c(depth, "depth")
    _f32c(pix_z, "pix_z")
    B, M, H, W = depth.shape
    V = pix_idx.shape[1]
    out = torch.empty(B, V, dtype=torch.float32, device=depth.device)
    check(lib.pm_tsdf_integrate_f32(_ptr(depth), _ptr(pix_idx), _ptr(pix_z), B, M, H * W, V, float(trunc),
                                    float(default_tsdf), _ptr(out), _stream()), "pm_tsdf_integrate_f32")
    return out


def depth_compact(world):
    """world (B,P,3) cropped cloud -> (compact (B,P,3): non-zero points + the first zero point, order kept; lengths (B) i32)."""
    _req(world)
    _f32c(world, "world")
    B, P, _ = world.shape
    out = torch.empty_like(world)
    lengths = torch.empty(B, dtype=torch.int32, device=world.device)
    check(lib.pm_depth_compact_f32(_ptr(world), B, P, _ptr(out), _ptr(lengths), _stream()), "pm_depth_compact_f32")
    return out, lengths


_HIP_RT = [None]


def cu_masked_stream(first, count, total=256):
    """A HIP stream whose kernels may only occupy CU-mask bits [first, first + count) (hipExtStreamCreateWithCUMask), as a
    torch.cuda.ExternalStream.  On the MI355X mask bit i is CU i // 8 of XCD i % 8 -- work-groups are still dealt round-robin over
    ALL eight XCDs, and a mask that leaves an XCD without a CU is ignored (tools/ubench/cu_mask_probe.hip,
    profiles/round6_cu_mask_probe.txt): a mask can split every XCD's CUs between two streams, it cannot give a stream its own
    XCDs / L2s.  hipGraphs replayed into the stream keep the mask.  Never destroyed (process lifetime)."""
    if _HIP_RT[0] is None:
        _HIP_RT[0] = C.CDLL("libamdhip64.so")
        _HIP_RT[0].hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        _HIP_RT[0].hipExtStreamCreateWithCUMask.restype = C.c_int
    words = (C.c_uint32 * ((total + 31) // 32))()
    for i in range(first, first + count):
        words[i // 32] |= 1 << (i % 32)
    h = C.c_void_p()
    rc = _HIP_RT[0].hipExtStreamCreateWithCUMask(C.byref(h), len(words), words)
    if rc != 0 or not h.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(h.value)


class FpsConfig(C.Structure):
    """pm_fps_config (include/partmanip_hip.h): the launch policy of the multi-work-group sampler, passed with every call."""
    _fields_ = [("max_groups", C.c_int), ("resident_cus", C.c_int), ("spin_limit", C.c_uint), ("spin_limit_set", C.c_int),
                ("legacy_shape", C.c_int)]


class _FpsPolicy:
    """Process-local policy behind fps_varlen's default (the LIBRARY has no state: this object is turned into a pm_fps_config
    for every call).  max_groups: cap on the work-groups per cloud (None: the library's own; 1: one work-group per cloud);
    resident_cus: the CUs a launch may occupy when the caller masks / shares the device (None: all); spin_limit: the hand-off
    poll budget (None: ~1 s; tests force give-ups with 0); legacy_shape: the round-2 launch shape (A/B).  The PM_FPS_MAXG /
    PM_FPS_SPIN_LIMIT / PM_FM_CFG environment variables of earlier rounds are read ONCE here, at import."""

    def __init__(self):
        e = os.environ
        self.max_groups = int(e["PM_FPS_MAXG"]) if "PM_FPS_MAXG" in e else None
        self.spin_limit = int(e["PM_FPS_SPIN_LIMIT"]) if "PM_FPS_SPIN_LIMIT" in e else None
        self.legacy_shape = e.get("PM_FM_CFG", "1") == "0"
        self.resident_cus = None
        self.gave_up_cap = False                              # set by the watch below: a give-up capped the group count

    def struct(self, max_groups=None, spin_limit=None, resident_cus=None):
        mg = self.max_groups if max_groups is None else max_groups
        sl = self.spin_limit if spin_limit is None else spin_limit
        rc = self.resident_cus if resident_cus is None else resident_cus
        return FpsConfig(-1 if mg is None else int(mg), 0 if rc is None else int(rc), 0 if sl is None else int(sl),
                         0 if sl is None else 1, int(self.legacy_shape))


FPS_POLICY = _FpsPolicy()


def fps_varlen(xyz, lengths, K, ws, pad=False, max_groups=None, spin_limit=None, resident_cus=None):
    """FPS over the first lengths[b] rows of each cloud of xyz (B, ld, D) -> idx (B, K) int32; pad=True: pytorch3d's
    -1 once a cloud is exhausted, pad=False: keep sampling (index 0 repeats), as the full depth cloud would.
    max_groups / spin_limit / resident_cus override FPS_POLICY for this call (see _FpsPolicy)."""
    _req(xyz, lengths)
    _f32c(xyz, "xyz")
    B, ld, Dd = xyz.shape
    idx = torch.empty(B, K, dtype=torch.int32, device=xyz.device)
    nb = int(lib.pm_fps_varlen_workspace_bytes(B, ld))
    w, base, al = None, 0, 0
    if nb:
        w = ws.get(nb + 8)
        base = w.data_ptr()
        al = (-base) % 8
    _fps_watch_poll(ws)
    cfg = FPS_POLICY.struct(max_groups, spin_limit, resident_cus)
    check(lib.pm_fps_varlen_cfg_f32(_ptr(xyz), B, ld, Dd, K, _ptr(lengths), int(pad), _ptr(idx), C.byref(cfg),
                                    (base + al) if w is not None else None, (w.numel() - al) if w is not None else 0, _stream()),
          "pm_fps_varlen_cfg_f32")
    # a multi-work-group launch that gives up degrades on the device (a launch queued behind it re-samples the big clouds when
    # the reservation's last word is set): nothing to check here, no host sync.  fps_varlen_gave_up(ws) reads the word; the
    # watch below copies it to pinned host memory behind the launch and the NEXT call looks at it (no sync either).  Only a call
    # that finds no earlier watch pending arms one (give-ups of the calls in between are seen by fps_varlen_gave_up only), and
    # none is armed inside a stream capture (the pinned copy and the event are not capturable work).
    multi = nb and int(lib.pm_fps_varlen_groups_cfg(B, ld, Dd, C.byref(cfg))) >= 2
    ws.fps_err = w[al + nb - 8: al + nb].view(torch.int64) if multi else None
    if ws.fps_err is not None and getattr(ws, "_fps_watch", None) is None and not torch.cuda.is_current_stream_capturing():
        host = torch.empty(1, dtype=torch.int64, pin_memory=True)
        host.copy_(ws.fps_err, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ws._fps_watch = (host, ev, cfg.spin_limit_set != 0)                        # (a forced budget -- a test: report, do not act)
    return idx


_FPS_GIVE_UPS = [0]


def _fps_watch_poll(ws):
    """A multi-work-group FPS launch that gave up costs its whole spin budget (~1 s) before the one-work-group sampler redoes the
    clouds -- correct, but silent.  The give-up word of an earlier call on this workspace, copied to the host behind that call:
    once it has arrived and is set, say so and cap FPS_POLICY.max_groups at 1 for the rest of the process (a module-level flag
    that every later call PASSES to the library -- nothing in the environment or the library changes; not when that call ran
    with a forced spin budget: the tests' way to provoke give-ups)."""
    watch = getattr(ws, "_fps_watch", None)
    if watch is None or not watch[1].query():
        return
    ws._fps_watch = None
    if int(watch[0][0]) != 0:
        _FPS_GIVE_UPS[0] += 1
        if not watch[2] and FPS_POLICY.max_groups != 1:
            import warnings
            warnings.warn("pm_fps_varlen_cfg_f32: a multi-work-group sampling launch gave up waiting for its partner work-groups (the "
                          "device is shared or CU-masked?) and fell back to one work-group per cloud; ops.FPS_POLICY.max_groups = 1 "
                          "from now on (pass resident_cus to keep the multi-work-group path on a masked device)")
            FPS_POLICY.max_groups = 1
            FPS_POLICY.gave_up_cap = True


def fps_varlen_gave_up(ws):
    """Diagnostics (a host read): did the last multi-work-group fps_varlen call on this workspace fall back to the
    one-work-group sampler because a work-group's partners were not resident?"""
    e = getattr(ws, "fps_err", None)
    return bool(e is not None and int(e.item()) != 0)


def tsdf_sparse_voxel(vol, K, lo, hi, ws):
    """vol (B, res, res, res) -> (B, K, 4) rows (x, y, z, tsdf) of FPS-sampled voxels with lo < tsdf < hi."""
    _req(vol)
    _f32c(vol, "vol")
    B, res = vol.shape[0], vol.shape[1]
    V = res ** 3
    coords = torch.empty(B, V, 3, dtype=torch.float32, device=vol.device)
    lengths = torch.empty(B, dtype=torch.int32, device=vol.device)
    check(lib.pm_tsdf_select_f32(_ptr(vol), B, res, float(lo), float(hi), _ptr(coords), _ptr(lengths), _stream()),
          "pm_tsdf_select_f32")
    idx = fps_varlen(coords, lengths, K, ws, pad=True)
    out = torch.empty(B, K, 4, dtype=torch.float32, device=vol.device)
    check(lib.pm_tsdf_sparse_gather_f32(_ptr(coords), _ptr(idx), _ptr(vol), B, res, K, _ptr(out), _stream()),
          "pm_tsdf_sparse_gather_f32")
    return out


def gaussian_sample(mu, log_std, eps, max_action, act_tanh, want_log_std_rows=True):
    """actor_critic.py:36-47 after the forwards: -> (squashed actions (B, A), logp (B,), log_std rows (B, A))."""
    _req(mu, log_std, eps)
    _f32c(eps, "eps")
    B, A = mu.shape
    actions = torch.empty(B, A, dtype=torch.float32, device=mu.device)
    logp = torch.empty(B, dtype=torch.float32, device=mu.device)
    rows = torch.empty(B, A, dtype=torch.float32, device=mu.device) if want_log_std_rows else None
    check(lib.pm_gaussian_sample_f32(_ptr(mu), _rows(mu, "mu"), _ptr(log_std), _ptr(eps), B, A, float(max_action),
                                     int(act_tanh), _ptr(actions), _ptr(logp), _ptr(rows), _stream()),
          "pm_gaussian_sample_f32")
    return actions, logp, rows


def rms_update(x, n_new, mean, S, std, ws):
    """RMS.py:10-18 for one (N, D) batch; mean / S / std ((1, D) fp32, contiguous) are updated in place."""
    _req(x, mean, S, std)
    N, D = x.shape
    w = ws.get(lib.pm_rms_update_workspace_bytes(D))
    check(lib.pm_rms_update_f32(_ptr(x), _rows(x, "x"), N, D, int(n_new), _ptr(mean), _ptr(S), _ptr(std), _ptr(w),
                                w.numel(), _stream()), "pm_rms_update_f32")


def rms_update_dp(x, n_new, mean, S, std, ws, mom, sum_over_ranks, world):
    """The same update when the batch is sharded over `world` ranks: local column moments -> `sum_over_ranks(mom)`
    (one all-reduce of 2*D doubles) -> RMS.py:10-18 from the global moments over N * world rows."""
    _req(x, mean, S, std, mom)
    N, D = x.shape
    w = ws.get(lib.pm_rms_update_workspace_bytes(D))
    check(lib.pm_rms_moments_f64(_ptr(x), _rows(x, "x"), N, D, _ptr(mom), _ptr(w), w.numel(), _stream()), "pm_rms_moments_f64")
    sum_over_ranks(mom)
    check(lib.pm_rms_apply_moments_f32(_ptr(mom), N * world, D, int(n_new), _ptr(mean), _ptr(S), _ptr(std), _stream()),
          "pm_rms_apply_moments_f32")


def rms_normalize(x, mean, std):
    _req(x, mean, std)
    N, D = x.shape
    out = torch.empty(N, D, dtype=torch.float32, device=x.device)
    check(lib.pm_rms_normalize_f32(_ptr(x), _rows(x, "x"), N, D, _ptr(mean), _ptr(std), _ptr(out), D, _stream()),
          "pm_rms_normalize_f32")
    return out
