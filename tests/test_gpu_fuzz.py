"""Seeded random-shape sweeps of the general-purpose kernels against plain torch CPU references: odd sizes,
unaligned leading dimensions, strided views -- the shapes the fixed-size parity tests do not visit."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ops():
    from partmanip_amd import ops as o
    return o


def rel(a, b):
    b = b.double()
    return float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


def test_linear_kernels_random_shapes():
    o = ops()
    rng = np.random.default_rng(0)
    ws = o.Workspace(torch.device(DEV))
    for it in range(40):
        M, K, N = int(rng.integers(1, 700)), int(rng.integers(1, 300)), int(rng.integers(1, 200))
        padx, padw = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        g = torch.Generator().manual_seed(it)
        xb = torch.randn(M, K + padx, generator=g)
        wb = torch.randn(N, K + padw, generator=g) / max(K, 1) ** 0.5
        x, w = xb[:, :K], wb[:, :K]
        b = torch.randn(N, generator=g)
        act = int(rng.integers(0, 2))
        y = torch.empty(M, N, device=DEV)
        xd, wd = xb.to(DEV)[:, :K], wb.to(DEV)[:, :K]
        o.linear_fwd(xd, wd, b.to(DEV), y, act)
        ref = x @ w.t() + b
        ref = torch.tanh(ref) if act else ref
        assert rel(y, ref) < 3e-5, (it, M, K, N)
        dy = torch.randn(M, N, generator=g)
        dx = torch.empty(M, K, device=DEV)
        h = torch.tanh(torch.randn(M, K, generator=g))
        o.linear_bwd_data(dy.to(DEV), wd, h.to(DEV) if act else None, dx, act)
        dref = dy @ w
        dref = dref * (1 - h * h) if act else dref
        assert rel(dx, dref) < 3e-5, (it, M, K, N)
        dw = torch.empty(N, K, device=DEV)
        db = torch.empty(N, device=DEV)
        o.linear_bwd_weight(dy.to(DEV), xd, dw, db, ws)
        assert rel(dw, dy.t() @ x) < 3e-5 and rel(db, dy.sum(0)) < 3e-5, (it, M, K, N)


def test_conv3d_gather_scatter_random_geometry():
    """pm_im2col3d_f32 + Linear == F.conv3d, and pm_col2im3d_f32 == its input gradient, for random (C, extent, k,
    stride, pad), channels-first and channels-last storage."""
    o = ops()
    rng = np.random.default_rng(1)
    for it in range(25):
        B, C, Co = int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(1, 9))
        D, H, W = (int(v) for v in rng.integers(3, 12, size=3))
        k = int(rng.choice([1, 2, 3, 5]))
        st, pad = int(rng.integers(1, 4)), int(rng.integers(0, k // 2 + 2))
        if min(D, H, W) + 2 * pad < k:
            continue
        g = torch.Generator().manual_seed(100 + it)
        x = torch.randn(B, C, D, H, W, generator=g, requires_grad=True)
        w = torch.randn(Co, C, k, k, k, generator=g) * 0.2
        ref = F.conv3d(x, w, stride=st, padding=pad)
        xd = x.detach().to(DEV)
        if it % 2:                                           # channels-last storage behind an NCDHW view
            xd = xd.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
        ldc = (C * k ** 3 + 3) // 4 * 4
        cols = o.im2col3d(xd, k, st, pad, ldc)
        y = torch.empty(cols.shape[0], Co, device=DEV)
        wp = torch.zeros(Co, ldc, device=DEV)
        wp[:, :C * k ** 3] = w.reshape(Co, -1).to(DEV)
        o.linear_fwd(cols, wp, None, y, o.ACT_NONE)
        got = y.view(B, *ref.shape[2:], Co).permute(0, 4, 1, 2, 3)
        assert rel(got, ref.detach()) < 3e-5, (it, B, C, D, H, W, k, st, pad)
        dyr = torch.randn(ref.shape, generator=g)
        (dx_ref,) = torch.autograd.grad((ref * dyr).sum(), x)
        dy2 = dyr.permute(0, 2, 3, 4, 1).reshape(-1, Co).contiguous().to(DEV)
        dcols = torch.empty_like(cols)
        o.linear_bwd_data(dy2, wp, None, dcols, o.ACT_NONE)
        dx = torch.empty_like(xd)
        o.col2im3d(dcols, dx, k, st, pad)
        assert rel(dx, dx_ref) < 3e-5, (it, B, C, D, H, W, k, st, pad)


def test_conv3d_input_layer_direct_kernels_random_geometry():
    """pm_conv3d_c1_fwd_f32 / pm_conv3d_c1_wgrad_f32 (the single-channel 5^3 x 16 input layer as a direct stencil) against
    F.conv3d and its autograd weight / bias gradients for random extents (D != H != W), strides, paddings, batch sizes
    (work-group slices that end mid-volume) and a strided (row-of-a-wider-tensor) input view."""
    o = ops()
    rng = np.random.default_rng(7)
    for it in range(12):
        B = int(rng.choice([1, 2, 5, 40]))
        D, H, W = (int(v) for v in rng.integers(5, 30, size=3))
        st, pad = int(rng.integers(1, 4)), int(rng.integers(0, 4))
        g = torch.Generator().manual_seed(700 + it)
        tail = int(rng.integers(0, 9))                       # the volume is the head of a wider observation row
        row = torch.randn(B, D * H * W + tail, generator=g)
        x = row[:, :D * H * W].unflatten(1, (1, D, H, W)).detach().clone().requires_grad_(True)
        w = (torch.randn(16, 1, 5, 5, 5, generator=g) * 0.1).requires_grad_(True)
        bias = (torch.randn(16, generator=g) * 0.1).requires_grad_(True)
        act = bool(it % 2)
        ref = F.conv3d(x, w, bias, stride=st, padding=pad)
        if act:
            ref = torch.tanh(ref)
        x5 = row.to(DEV)[:, :D * H * W].unflatten(1, (1, D, H, W))
        wt = w.detach().reshape(16, 125).t().contiguous().to(DEV)
        y = o.conv3d_c1_fwd(x5, 5, st, pad, wt, bias.detach().to(DEV), o.ACT_TANH if act else o.ACT_NONE)
        got = y.view(B, *ref.shape[2:], 16).permute(0, 4, 1, 2, 3)
        assert rel(got, ref.detach()) < 3e-5, (it, B, D, H, W, st, pad)
        dyr = torch.randn(ref.shape, generator=g)
        dw_ref, db_ref = torch.autograd.grad((ref * dyr).sum(), [w, bias])
        dz = dyr * (1 - ref.detach() ** 2) if act else dyr    # gradient at the pre-activation
        dz2 = dz.permute(0, 2, 3, 4, 1).reshape(-1, 16).contiguous().to(DEV)
        dw = torch.empty(16, 125, device=DEV)
        db = torch.empty(16, device=DEV)
        o.conv3d_c1_wgrad(dz2, x5, 5, st, pad, dw, db, o.Workspace(DEV))
        assert rel(dw, dw_ref.reshape(16, 125)) < 5e-5, (it, "dW", B, D, H, W, st, pad)
        assert rel(db, db_ref) < 5e-5, (it, "db")


def test_gae_random_shapes_bit_exact():
    o = ops()
    rng = np.random.default_rng(2)
    for it in range(20):
        T, N = int(rng.integers(1, 40)), int(rng.integers(1, 300))
        g = torch.Generator().manual_seed(200 + it)
        r, v = torch.randn(T, N, 1, generator=g), torch.randn(T, N, 1, generator=g)
        d = torch.rand(T, N, 1, generator=g) < 0.1
        s = d & (torch.rand(T, N, 1, generator=g) < 0.5)
        lv = torch.randn(N, 1, generator=g)
        succ = [None, 500.0][it % 2]
        ret_ref, adv_ref = R.gae_returns(r, v, d, s, lv, 0.99, 0.95, succ, False)
        ret, adv = torch.empty(T, N, 1, device=DEV), torch.empty(T, N, 1, device=DEV)
        o.gae_scan(r.to(DEV), v.to(DEV), d.to(DEV), s.to(DEV), lv.to(DEV), ret, adv, 0.99, 0.95, succ)
        assert torch.equal(ret.cpu(), ret_ref) and torch.equal(adv.cpu(), adv_ref), (it, T, N)


def test_ball_query_and_fps_random_shapes_bit_exact():
    o = ops()
    rng = np.random.default_rng(3)
    ws = o.Workspace(torch.device(DEV))
    for it in range(16):
        B, P = int(rng.integers(1, 6)), int(rng.integers(8, 1500))
        K, ns = int(rng.integers(1, min(P, 200) + 1)), int(rng.choice([4, 16, 32]))
        g = torch.Generator().manual_seed(300 + it)
        xyz = torch.rand(B, P, 3, generator=g) * 2 - 1
        idx = o.fps(xyz.to(DEV), K, ws).cpu().numpy()
        assert np.array_equal(idx, R.fps(xyz.numpy(), K).astype(np.int32)), (it, B, P, K)
        ctr = torch.gather(xyz, 1, torch.from_numpy(idx).long().unsqueeze(-1).expand(B, K, 3)).contiguous()
        radius = float(rng.uniform(0.05, 0.6))
        got = o.ball_query(xyz.to(DEV), ctr.to(DEV), radius, ns).cpu().numpy()
        assert np.array_equal(got, R.ball_query(xyz.numpy(), ctr.numpy(), radius, ns)), (it, B, P, K, ns)


def test_pointnet_encoder_random_configs():
    """Fused encoder forward + backward (saved layer 2 and recompute) over random (B, points, channels, pooling,
    centring, proprio) against the CPU restatement with the pooling index pinned."""
    from partmanip_amd.algo_utils import ActorCritic
    rng = np.random.default_rng(4)
    # plus fixed corner cases: more clouds than CUs (persistent work-groups take several clouds each, 300 and 515 are
    # not multiples of the grid), the largest cloud the backward's key tables hold (4096 points), one cloud
    corners = {10: (300, 64, 3), 11: (515, 128, 4), 12: (2, 4096, 3), 13: (1, 2048, 5)}
    for it in range(14):
        B, P, C = int(rng.integers(1, 9)), int(rng.choice([64, 128, 320, 1024, 1536])), int(rng.integers(3, 9))
        max_mean, sub_mean, proprio = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.choice([0, 3]))
        if it in corners:
            B, P, C = corners[it]
        net = dict(name="PointNet", activation="tanh", max_mean=max_mean, sub_mean=sub_mean, point_num=P,
                   save_h2=bool(it % 2))
        torch.manual_seed(400 + it)
        ac = ActorCritic(P * C + proprio, 6, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net),
                         proprio).to(DEV)
        f = ac.flat()
        g = torch.Generator().manual_seed(500 + it)
        x = torch.cat([(torch.rand(B, P, C, generator=g) * 2 - 1).reshape(B, -1), torch.randn(B, proprio, generator=g)], 1)
        p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ac.state_dict().items()}
        out = ac.actor.hip_forward(x.to(DEV))
        am = ac.actor._saved[2].cpu().long()
        ref = R.pointnet_forward(p, "actor", net, x.clone(), proprio, point_num=P, argmax_override=am)
        assert rel(out, ref.detach()) < 3e-5, (it, B, P, C)
        dy = torch.randn(B, 6, generator=g)
        names = [k for k in p if k.startswith("actor.")]
        grads = torch.autograd.grad((ref * dy).sum(), [p[k] for k in names])
        ac.actor.hip_backward(dy.to(DEV))
        off = 0
        for k, v in ac.actor.named_parameters():
            got = f["grad_actor"][off:off + v.numel()].view(v.shape)
            off += v.numel()
            assert rel(got, grads[names.index("actor." + k)]) < 3e-4, (it, B, P, C, k)


def test_pointnet2_random_level_geometry():
    """PointNet2 with the fused set-abstraction shapes over random (B, centres per level, radii, input channels,
    proprio): sampled / grouped indices bit-exact, outputs and gradients against the CPU restatement (pooling pinned)."""
    from partmanip_amd.algo_utils import ActorCritic
    rng = np.random.default_rng(5)
    for it in range(5):
        B, C = int(rng.integers(1, 4)), int(rng.choice([3, 4, 6]))
        n0 = int(rng.integers(20, 140))
        n1 = int(rng.integers(3, min(40, n0) + 1))
        proprio = int(rng.choice([0, 4]))
        net = dict(name="PointNet2", activation="tanh", npoints=[n0, n1], radii=[float(rng.uniform(0.15, 0.4)), float(rng.uniform(0.3, 0.8))],
                   nsamples=[32, 32], mlps=[[64, 64, 128], [128, 128, 256], [256, 512]], save_h2=bool(it % 2))
        torch.manual_seed(600 + it)
        ac = ActorCritic(1024 * C + proprio, 5, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net),
                         proprio).to(DEV)
        f = ac.flat()
        assert ac.actor._fused == [True, True]
        g = torch.Generator().manual_seed(700 + it)
        x = torch.cat([(torch.rand(B, 1024, C, generator=g) * 2 - 1).reshape(B, -1), torch.randn(B, proprio, generator=g)], 1).contiguous()
        p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ac.state_dict().items()}
        out_ref, aux, _ = R.pointnet2_forward(p, "actor", net, x.clone(), proprio, return_aux=True)
        out = ac.actor.hip_forward(x.to(DEV))
        saved = ac.actor._saved
        for l, (_, idx_g) in enumerate(aux):
            assert torch.equal(saved[l][0].cpu().long(), idx_g), (it, l)
        assert rel(out, out_ref.detach()) < 5e-5, it
        pin = R.pointnet2_forward(p, "actor", net, x.clone(), proprio, pool_args=[s_[1].cpu().long() for s_ in saved])
        dy = torch.randn(B, 5, generator=g)
        names = [k for k in p if k.startswith("actor.")]
        grads = torch.autograd.grad((pin * dy).sum(), [p[k] for k in names])
        ac.actor.hip_backward(dy.to(DEV))
        off = 0
        for k, v in ac.actor.named_parameters():
            got = f["grad_actor"][off:off + v.numel()].view(v.shape)
            off += v.numel()
            assert rel(got, grads[names.index("actor." + k)]) < 3e-4, (it, k)


def test_loss_kernels_random_sizes():
    """Value loss and the DAgger / BC MSE loss (all three target modes) over random sizes against plain torch."""
    o = ops()
    rng = np.random.default_rng(6)
    for it in range(12):
        B, A = int(rng.integers(1, 3000)), int(rng.integers(1, 12))
        g = torch.Generator().manual_seed(800 + it)
        v, ret, old = (torch.randn(B, 1, generator=g) for _ in range(3))
        scal, dv = torch.zeros(8, device=DEV), torch.empty(B, 1, device=DEV)
        clipped = bool(it % 2)
        o.value_loss(v.to(DEV), ret.to(DEV), old.to(DEV), clipped, 0.2, None, 1.0, scal, dv)
        vv = v.clone().requires_grad_(True)
        loss = R.value_loss_fn(vv, ret, old, 0.2, clipped)
        (gv,) = torch.autograd.grad(loss, vv)
        assert abs(float(scal[0]) - float(loss)) <= 2e-5 * max(1.0, abs(float(loss))) and rel(dv, gv) < 2e-5, (it, B)
        stu, tea = torch.randn(B, A, generator=g), torch.randn(B, A, generator=g)
        for mode in (1, 3, 0):
            d = torch.empty(B, A, device=DEV)
            o.mse_tanh_loss(stu.to(DEV), tea.to(DEV), 1.0, mode, 1.0, scal, d)
            s_ = stu.clone().requires_grad_(True)
            sa = torch.tanh(s_) if mode & 1 else s_
            ta = tea if (mode & 2 or not mode & 1) else torch.tanh(tea)
            l2 = (ta - sa).pow(2).mean()
            (gs,) = torch.autograd.grad(l2, s_)
            assert abs(float(scal[0]) - float(l2)) <= 3e-5 * max(1e-3, float(l2)) and rel(d, gs) < 3e-5, (it, mode)
