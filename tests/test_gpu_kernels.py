"""Parity of each HIP kernel (called through the C ABI via partmanip_amd.ops) against the CPU
oracle on identical seeded inputs.  GPU box only (`-m gpu`).

Tolerances (fp32 path; the oracle runs torch CPU fp32, kernels use exact-fp32 MFMA with a
different summation order):
  GAE returns/advantages ............ bit-exact
  whole-batch normalised advantages . rtol 2e-6 (mean/std reduced in fp64 on both sides)
  Linear / encoder outputs .......... atol 2e-5 + rtol 2e-5 of the output scale
  gradients ......................... 1e-4 relative to the gradient tensor's max-abs
  integer outputs (argmax on well-separated inputs, FPS, ball query) bit-exact
"""
import math

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.golden import cases
from tests.helpers import load_fixture, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ops():
    from partmanip_amd import ops as _ops
    return _ops


def rel_err(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))


# ------------------------------------------------------------------------------- GAE
@pytest.mark.parametrize("name", list(cases.GAE_CASES))
def test_gae_golden_bit_exact(name):
    c, fx = cases.GAE_CASES[name], load_fixture(name)
    inp = cases.gae_inputs(c)
    from partmanip_amd.algo_utils import RolloutStorage
    st = RolloutStorage(c["N"], c["T"], 3, 2, DEV, c["succ_value"], c["whole_adv_norm"])
    st.rewards.copy_(t(inp["rewards"]))
    st.values.copy_(t(inp["values"]))
    st.dones.copy_(t(inp["dones"]))
    st.succs.copy_(t(inp["succs"]))
    st.compute_returns(t(inp["last_values"]).to(DEV), 0.99, 0.95)
    assert np.array_equal(st.returns.cpu().numpy(), fx["returns"])
    if c["whole_adv_norm"]:
        np.testing.assert_allclose(st.advantages.cpu().numpy(), fx["advantages"], rtol=2e-6, atol=2e-7)
    else:
        assert np.array_equal(st.advantages.cpu().numpy(), fx["advantages"])


# (300, 37) / (129, 16) / (257, 4100): several 128-step chunks with the chain state carried across them, ragged last chunk,
# env counts that are not a multiple of the 16 envs a work-group owns
@pytest.mark.parametrize("T,N,succ", [(128, 4096, None), (8, 4096, 500.0), (64, 256, 0), (3, 1, None), (5, 67, 2.5),
                                      (300, 37, None), (129, 16, 3.0), (257, 4100, None), (1, 5, 1.0)])
def test_gae_full_size_vs_oracle(T, N, succ):
    g = torch.Generator().manual_seed(T * 1000 + N)
    rewards, values = torch.randn(T, N, 1, generator=g), torch.randn(T, N, 1, generator=g)
    dones = torch.rand(T, N, 1, generator=g) < 0.02
    succs = dones & (torch.rand(T, N, 1, generator=g) < 0.5)
    last = torch.randn(N, 1, generator=g)
    ret_ref, adv_ref = R.gae_returns(rewards, values, dones, succs, last, 0.99, 0.95, succ, False)
    ret, adv = torch.empty(T, N, 1, device=DEV), torch.empty(T, N, 1, device=DEV)
    ops().gae_scan(rewards.to(DEV), values.to(DEV), dones.to(DEV), succs.to(DEV), last.to(DEV), ret, adv, 0.99, 0.95, succ)
    assert torch.equal(ret.cpu(), ret_ref)
    assert torch.equal(adv.cpu(), adv_ref)


def test_moments_normalize_and_gather():
    o = ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(524288, generator=g) * 3 + 1.5
    xd = x.to(DEV)
    mom = torch.zeros(2, dtype=torch.float64, device=DEV)
    ws = o.Workspace(torch.device(DEV))
    o.moments(xd, mom, ws)
    np.testing.assert_allclose(mom.cpu().numpy(), [x.double().sum(), (x.double() ** 2).sum()], rtol=1e-12)
    o.normalize_apply(xd, mom, x.numel())
    ref = (x - x.mean()) / (x.std() + 1e-8)
    np.testing.assert_allclose(xd.cpu().numpy(), ref.numpy(), rtol=2e-6, atol=2e-6)
    for cols in (53, 64, 3072):
        src = torch.randn(1000, cols, generator=g)
        idx = torch.randperm(1000, generator=g)[:257]
        dst = torch.empty(257, cols, device=DEV)
        o.gather_rows(src.to(DEV), idx.to(DEV), dst)
        assert torch.equal(dst.cpu(), src[idx])


# ------------------------------------------------------------------------------- Linear
@pytest.mark.parametrize("M,N,K,act", [(2048, 512, 53, 1), (2048, 512, 512, 1), (2048, 10, 512, 0), (2048, 1, 512, 0),
                                       (63, 33, 19, 1), (15, 64, 64, 1), (2048, 128, 1031, 1), (300, 32, 128, 0),
                                       # weight gradients of <= 32 rows: the LDS-DMA kernel's 32 x 128 tile (ragged in N, K, M)
                                       (4096, 32, 864, 0), (2048, 16, 200, 1), (70000, 32, 1728, 0), (1000, 28, 132, 0), (2048, 32, 64, 0)])
def test_linear_fwd_bwd(M, N, K, act):
    o = ops()
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    h_in = torch.tanh(torch.randn(M, K, generator=g))         # stands for "x is a tanh output" in bwd_data
    xd, wd, bd, dyd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
    y = torch.empty(M, N, device=DEV)
    o.linear_fwd(xd, wd, bd, y, act)
    z = x.double() @ w.double().T + b.double()
    ref = torch.tanh(z) if act else z
    assert rel_err(y, ref) < 2e-5
    dx = torch.empty(M, K, device=DEV)
    o.linear_bwd_data(dyd, wd, h_in.to(DEV) if act else None, dx, act)
    dref = dy.double() @ w.double()
    if act:
        dref = dref * (1 - h_in.double() ** 2)
    assert rel_err(dx, dref) < 2e-5
    dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    o.linear_bwd_weight(dyd, xd, dw, db, o.Workspace(torch.device(DEV)))
    assert rel_err(dw, dy.double().T @ x.double()) < 2e-5
    assert rel_err(db, dy.double().sum(0)) < 2e-5


def test_linear_strided_views():
    """Operands that are column slices of wider buffers (row stride > width), as the PointNet head uses."""
    o = ops()
    g = torch.Generator().manual_seed(9)
    big = torch.randn(100, 1031, generator=g)
    w, b = torch.randn(128, 1031, generator=g) / 32, torch.randn(128, generator=g)
    x = big.to(DEV)
    y = torch.empty(100, 200, device=DEV)
    o.linear_fwd(x, w.to(DEV), b.to(DEV), y[:, 8:136], 1)
    ref = torch.tanh(big.double() @ w.double().T + b.double())
    assert rel_err(y[:, 8:136], ref) < 2e-5


# ------------------------------------------------------------------------------- losses
@pytest.mark.parametrize("B,A,mini_norm", [(2048, 10, False), (2048, 10, True), (37, 7, True), (1500, 3, False), (40000, 10, False)])
def test_ppo_actor_loss(B, A, mini_norm):
    o = ops()
    g = torch.Generator().manual_seed(B + A)
    mu = (torch.randn(B, A, generator=g) * 0.3).requires_grad_(True)
    log_std = (torch.full((A,), math.log(0.5)) + 0.1 * torch.randn(A, generator=g)).requires_grad_(True)
    actions = torch.rand(B, A, generator=g) * 1.998 - 0.999
    actions.view(-1)[::13] = 1.0
    adv = torch.randn(B, 1, generator=g)
    old_mu = mu.detach() + 0.05 * torch.randn(B, A, generator=g)
    old_sigma = log_std.detach().repeat(B, 1) + 0.02 * torch.randn(B, A, generator=g)
    x = R.action_deactivation(actions, "tanh", 1.0)
    logp, ent = R.gaussian_logp_entropy(mu, log_std, x)
    old_logp = (logp.detach() + 0.3 * torch.randn(B, generator=g)).view(B, 1)     # wide ratios: clip branches hit
    kl_ref, loss_ref = R.actor_loss_terms(logp, mu, log_std.repeat(B, 1), old_logp, adv, old_mu, old_sigma, 0.2, mini_norm)
    gmu, gls = torch.autograd.grad(loss_ref, [mu, log_std])

    d = lambda v: v.detach().to(DEV).contiguous()
    scal, dmu, dls = torch.zeros(8, device=DEV), torch.empty(B, A, device=DEV), torch.empty(A, device=DEV)
    mom, cnt = None, 0
    if mini_norm:
        mom = torch.zeros(2, dtype=torch.float64, device=DEV)
        o.moments(d(adv).view(-1), mom, o.Workspace(torch.device(DEV)))
        cnt = B
    o.ppo_actor_loss(d(mu), d(log_std), d(actions), d(old_logp), d(adv), d(old_mu), d(old_sigma), 1.0, True, 0.2, 0.1,
                     mom, cnt, scal, dmu, dls, o.Workspace(torch.device(DEV)))
    s = scal.cpu()
    np.testing.assert_allclose(float(s[0]), float(loss_ref.detach()), rtol=5e-6, atol=5e-6)   # mean of +-O(1) terms
    np.testing.assert_allclose(float(s[1]), float(kl_ref), rtol=1e-5, atol=2e-7)
    assert float(s[2]) == float(float(kl_ref) > 0.1)
    np.testing.assert_allclose(float(s[3]), float(ent[0]), rtol=5e-6, atol=5e-6)
    # saturated actions give |logp| ~ 1e2: one fp32 ulp there is ~1e-5 in the exponent of ratio = exp(logp - old)
    assert rel_err(dmu, gmu) < 2e-4
    assert rel_err(dls, gls) < 2e-4
    lp, en = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    o.gaussian_logp(d(mu), d(log_std), d(actions), 1.0, True, lp, en)
    np.testing.assert_allclose(lp.cpu().numpy(), logp.detach().numpy(), rtol=5e-6, atol=5e-5)


@pytest.mark.parametrize("B,K,A,mini_norm,hact", [(2048, 512, 10, False, "tanh"), (2048, 512, 10, True, "tanh"), (300, 64, 3, False, "elu"),
                                                  (4100, 128, 16, True, "tanh"), (2048, 512, 1, False, "relu"), (256, 1024, 16, False, "selu")])
def test_ppo_actor_head_equals_the_separate_launches(B, K, A, mini_norm, hact):
    """pm_ppo_actor_head_f32 (head forward + loss + dmu + head data gradient + last-work-group reduction in one launch) against
    pm_linear_fwd_f32 -> pm_ppo_actor_loss_fwd_bwd_f32 -> pm_linear_bwd_data_f32, which test_ppo_actor_loss / test_linear_* pin
    to the oracle: every output bit for bit (B >= 256: the row-wise Linear kernels on both sides), run three times on one
    workspace (the work-group counter must come back to zero)."""
    o = ops()
    g = torch.Generator().manual_seed(B + K + A)
    d = lambda v: v.to(DEV).contiguous()
    HACT = {"tanh": o.ACT_TANH, "relu": o.ACT_RELU, "elu": o.ACT_ELU, "selu": o.ACT_SELU}[hact]     # hidden activation: dH's epilogue
    h = d(torch.tanh(torch.randn(B, K, generator=g)))
    W, b = d(torch.randn(A, K, generator=g) * 0.05), d(torch.randn(A, generator=g) * 0.1)
    ls = d(torch.full((A,), math.log(0.5)) + 0.1 * torch.randn(A, generator=g))
    act = torch.rand(B, A, generator=g) * 1.998 - 0.999
    act.view(-1)[::13] = 1.0
    act, olp, adv = d(act), d(torch.randn(B, generator=g) * 0.3 - 4.0), d(torch.randn(B, generator=g))
    om, osg = d(torch.randn(B, A, generator=g) * 0.1), d(ls.cpu().repeat(B, 1) + 0.02 * torch.randn(B, A, generator=g))
    mom, cnt = None, 0
    if mini_norm:
        mom = torch.zeros(2, dtype=torch.float64, device=DEV)
        o.moments(adv, mom, o.Workspace(torch.device(DEV)))
        cnt = B
    mu, dmu, dh = torch.empty(B, A, device=DEV), torch.empty(B, A, device=DEV), torch.empty_like(h)
    scal, dls = torch.zeros(8, device=DEV), torch.empty(A, device=DEV)
    o.linear_fwd(h, W, b, mu, o.ACT_NONE)
    o.ppo_actor_loss(mu, ls, act, olp, adv, om, osg, 1.0, True, 0.2, 0.016, mom, cnt, scal, dmu, dls, o.Workspace(torch.device(DEV)))
    o.linear_bwd_data(dmu, W, h, dh, HACT)
    mu2, dmu2, dh2 = torch.empty_like(mu), torch.empty_like(dmu), torch.empty_like(dh)
    scal2, dls2 = torch.zeros(8, device=DEV), torch.empty(A, device=DEV)
    assert o.ppo_actor_head_supported(h, W, dh2)
    ws = o.Workspace(torch.device(DEV))
    for _ in range(3):
        dh2.fill_(float("nan"))
        o.ppo_actor_head(h, W, b, HACT, ls, act, olp, adv, om, osg, 1.0, True, 0.2, 0.016, mom, cnt, scal2, dmu2, dh2, dls2, ws,
                         mu_out=mu2)
        assert int(ws.counter[0]) == 0
        for name, x, y in (("mu", mu, mu2), ("dmu", dmu, dmu2), ("dh", dh, dh2), ("scal", scal, scal2), ("dlog_std", dls, dls2)):
            assert torch.equal(x, y), name
    assert not o.ppo_actor_head_supported(h[:, :K - 2], W[:, :K - 2], dh2[:, :K - 2])       # K % 4 != 0: the separate kernels


@pytest.mark.parametrize("B,K,clipped", [(2048, 512, False), (2048, 512, True), (300, 64, True), (4100, 128, False), (256, 1024, True)])
def test_value_head_equals_the_separate_launches(B, K, clipped):
    """pm_value_head_f32 against pm_linear_fwd_f32 -> pm_value_loss_fwd_bwd_f32 -> pm_linear_bwd_data_f32 (pinned to the oracle by
    test_value_loss / test_linear_*): V, dV, dH and the clip width bit for bit, the loss scalar to double-precision rounding."""
    o = ops()
    g = torch.Generator().manual_seed(B + K)
    d = lambda v: v.to(DEV).contiguous()
    h = d(torch.tanh(torch.randn(B, K, generator=g)))
    W, b = d(torch.randn(1, K, generator=g) * 0.05), d(torch.randn(1, generator=g) * 0.1)
    ret, old = d(torch.randn(B, 1, generator=g)), d(torch.randn(B, 1, generator=g))
    v, dv, dh, scal = torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV), torch.empty_like(h), torch.zeros(8, device=DEV)
    o.linear_fwd(h, W, b, v, o.ACT_NONE)
    o.value_loss(v, ret, old, clipped, 0.2, None, 1.0, scal, dv)
    o.linear_bwd_data(dv, W, h, dh, o.ACT_TANH)
    v2, dh2, scal2 = torch.empty(B, device=DEV), torch.empty_like(h), torch.zeros(8, device=DEV)
    dv2 = o.padded_cols(B, 1, torch.device(DEV))                               # (B, 1) rows of stride 4, as the learner passes
    assert o.value_head_supported(h, W, dh2)
    ws = o.Workspace(torch.device(DEV))
    for _ in range(3):
        dh2.fill_(float("nan"))
        o.value_head(h, W, b, o.ACT_TANH, ret, old, clipped, 0.2, None, 1.0, scal2, dv2, dh2, ws, v_out=v2)
        assert int(ws.counter[0]) == 0
        assert torch.equal(v.view(-1), v2) and torch.equal(dv, dv2) and torch.equal(dh, dh2)
        assert float(scal[1]) == float(scal2[1])
        np.testing.assert_allclose(float(scal2[0]), float(scal[0]), rtol=1e-6)


@pytest.mark.parametrize("M,O,H,act", [(2048, 53, 512, "tanh"), (2048, 64, 512, "relu"), (1000, 37, 256, "tanh"), (2048, 53, 128, "elu")])
def test_chained_hidden_layers_equal_the_layer_by_layer_launches(M, O, H, act):
    """pm_linear_fwd_chain_f32 / pm_linear_bwd_data_chain_f32 (the hidden layers of one network in ONE launch: the work-groups of
    a 64-row stripe hand their tiles over inside the launch -- write-through stores, a stripe counter, sc1 loads) against the
    single-layer entry points, which test_linear_* pin to the oracle: every output BIT FOR BIT (same MFMA order), 40 launches
    whose inputs change every time (a stale tile of the previous launch would be a wrong sum) beside a copy on another stream
    (uneven load), no spin ever giving up."""
    o = ops()
    ACT = {"tanh": o.ACT_TANH, "relu": o.ACT_RELU, "elu": o.ACT_ELU}[act]
    g = torch.Generator().manual_seed(M + O + H)
    d = lambda v: v.to(DEV).contiguous()
    dev = torch.device(DEV)
    x0 = d(torch.randn(M, O, generator=g))
    Ws = [d(torch.randn(H, O, generator=g) / O ** 0.5), d(torch.randn(H, H, generator=g) / H ** 0.5), d(torch.randn(H, H, generator=g) / H ** 0.5)]
    bs = [d(torch.randn(H, generator=g) * 0.1) for _ in range(3)]
    dz0 = d(torch.randn(M, H, generator=g))
    ws = o.Workspace(dev)
    side = torch.cuda.Stream()
    big_a, big_b = torch.empty(64 << 20, device=DEV), torch.empty(64 << 20, device=DEV)
    for it in range(40):
        x = x0 * (1.0 + 0.02 * it)
        dz = dz0 * (1.0 - 0.01 * it)
        ref, cur = [], x
        for W, b in zip(Ws, bs):
            y = torch.empty(M, H, device=DEV)
            o.linear_fwd(cur, W, b, y, ACT)
            ref.append(y)
            cur = y
        dref, dcur = [], dz
        for i in (2, 1):
            dx = torch.empty(M, H, device=DEV)
            o.linear_bwd_data(dcur, Ws[i], ref[i - 1], dx, ACT)
            dref.append(dx)
            dcur = dx
        if it % 3 != 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                big_b.copy_(big_a)
        ys = [torch.full((M, H), float("nan"), device=DEV) for _ in range(3)]
        ins = [x] + ys[:-1]
        ok = o.linear_fwd_chain([(i_, W, b, y, ACT) for i_, W, b, y in zip(ins, Ws, bs, ys)], ws)
        assert ok == (M * H <= 256 * 64 * 64 and H % 64 == 0), "shape support changed"
        if not ok:
            return
        dxs = [torch.full((M, H), float("nan"), device=DEV) for _ in range(2)]
        items, dcur = [], dz
        for i, dx in zip((2, 1), dxs):
            items.append((dcur, Ws[i], ys[i - 1], dx, ACT))
            dcur = dx
        assert o.linear_bwd_data_chain(items, ws)
        for k in range(3):
            assert torch.equal(ys[k], ref[k]), (it, "forward layer", k)
        for k in range(2):
            assert torch.equal(dxs[k], dref[k]), (it, "data gradient", k)
        torch.cuda.current_stream().wait_stream(side)
    assert not o.chain_gave_up(ws)


def test_fused_heads_last_work_group_reduction_under_uneven_load():
    """The fused head launches hand their per-work-group partials to the LAST work-group to arrive INSIDE the launch (relaxed
    agent-scope atomic stores -> s_waitcnt vmcnt(0) -> a relaxed agent-scope counter RMW -> agent-scope atomic loads: the
    '8-byte agent atomics on both sides' form of MI355X_MICROARCH.md, which bypasses the non-coherent L1 / per-XCD L2 without a
    fence).  That relies on how gfx950 lowers sc1 atomics, so it is stressed here the way the guide asks: the full 128-work-group
    grid, 200 launches whose inputs CHANGE every launch (a stale partial of the previous launch would be a wrong sum), a second
    stream keeping part of the chip busy with a copy (uneven load), every scalar and log_std gradient bit for bit against the
    two-launch path each time."""
    o = ops()
    B, K, A = 2048, 512, 10
    g = torch.Generator().manual_seed(77)
    d = lambda v: v.to(DEV).contiguous()
    h = d(torch.tanh(torch.randn(B, K, generator=g)))
    W, b = d(torch.randn(A, K, generator=g) * 0.05), d(torch.randn(A, generator=g) * 0.1)
    Wv, bv = d(torch.randn(1, K, generator=g) * 0.05), d(torch.randn(1, generator=g) * 0.1)
    ls = d(torch.full((A,), math.log(0.5)) + 0.1 * torch.randn(A, generator=g))
    act = d(torch.rand(B, A, generator=g) * 1.998 - 0.999)
    olp, adv0 = d(torch.randn(B, generator=g) * 0.3 - 4.0), d(torch.randn(B, generator=g))
    om, osg = d(torch.randn(B, A, generator=g) * 0.1), d(ls.cpu().repeat(B, 1) + 0.02 * torch.randn(B, A, generator=g))
    ret0, old = d(torch.randn(B, 1, generator=g)), d(torch.randn(B, 1, generator=g))
    dev = torch.device(DEV)
    mu, dmu, dh = torch.empty(B, A, device=DEV), torch.empty(B, A, device=DEV), torch.empty_like(h)
    scal, dls = torch.zeros(8, device=DEV), torch.empty(A, device=DEV)
    dmu2, dh2, scal2, dls2 = torch.empty_like(dmu), torch.empty_like(dh), torch.zeros(8, device=DEV), torch.empty(A, device=DEV)
    v, dv, dhv, scv = torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV), torch.empty_like(h), torch.zeros(8, device=DEV)
    dv2, dhv2, scv2 = o.padded_cols(B, 1, dev), torch.empty_like(h), torch.zeros(8, device=DEV)
    ws_ref, ws_a, ws_v = o.Workspace(dev), o.Workspace(dev), o.Workspace(dev)
    side = torch.cuda.Stream()
    big_a, big_b = torch.empty(64 << 20, device=DEV), torch.empty(64 << 20, device=DEV)
    bad = 0
    for it in range(200):
        adv = adv0 * (1.0 + 0.01 * it) + 0.001 * it
        ret = ret0 + 0.01 * it
        o.linear_fwd(h, W, b, mu, o.ACT_NONE)
        o.ppo_actor_loss(mu, ls, act, olp, adv, om, osg, 1.0, True, 0.2, 0.016, None, 0, scal, dmu, dls, ws_ref)
        o.linear_fwd(h, Wv, bv, v, o.ACT_NONE)
        o.value_loss(v, ret, old, True, 0.2, None, 1.0, scv, dv)
        if it % 3 != 2:                                       # two launches in three run next to a 256 MB copy on another stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                big_b.copy_(big_a)
        o.ppo_actor_head(h, W, b, o.ACT_TANH, ls, act, olp, adv, om, osg, 1.0, True, 0.2, 0.016, None, 0, scal2, dmu2, dh2, dls2, ws_a)
        o.value_head(h, Wv, bv, o.ACT_TANH, ret, old, True, 0.2, None, 1.0, scv2, dv2, dhv2, ws_v)
        ok = torch.equal(scal, scal2) and torch.equal(dls, dls2) and torch.equal(dmu, dmu2) and float(scv[1]) == float(scv2[1]) \
            and abs(float(scv2[0]) - float(scv[0])) <= 1e-6 * abs(float(scv[0])) and torch.equal(dv, dv2)
        bad += 0 if ok else 1
        torch.cuda.current_stream().wait_stream(side)
    assert bad == 0, f"{bad} of 200 launches read stale / partial sums"
    assert int(ws_a.counter[0]) == 0 and int(ws_v.counter[0]) == 0


@pytest.mark.parametrize("B,clipped", [(2048, False), (2048, True), (15, True)])
def test_value_loss(B, clipped):
    o = ops()
    g = torch.Generator().manual_seed(B)
    v = torch.randn(B, 1, generator=g).requires_grad_(True)
    ret, old = torch.randn(B, 1, generator=g) * 2, torch.randn(B, 1, generator=g)
    loss = R.value_loss_fn(v, ret, old, 0.2, clipped)
    gv, = torch.autograd.grad(loss, [v])
    scal, dv = torch.zeros(8, device=DEV), torch.empty(B, 1, device=DEV)
    o.value_loss(v.detach().to(DEV), ret.to(DEV), old.to(DEV), clipped, 0.2, None, 1.0, scal, dv)
    np.testing.assert_allclose(float(scal[0]), float(loss), rtol=2e-6)
    assert rel_err(dv, gv) < 1e-5


@pytest.mark.parametrize("B,A", [(2048, 10), (33, 7)])
def test_mse_tanh_loss(B, A):
    o = ops()
    g = torch.Generator().manual_seed(B)
    sm = torch.randn(B, A, generator=g).requires_grad_(True)
    tm = torch.randn(B, A, generator=g)
    loss = (torch.tanh(tm) * 1.0 - torch.tanh(sm) * 1.0).pow(2).mean()
    gs, = torch.autograd.grad(loss, [sm])
    scal, ds = torch.zeros(8, device=DEV), torch.empty(B, A, device=DEV)
    o.mse_tanh_loss(sm.detach().to(DEV), tm.to(DEV), 1.0, True, 1.0, scal, ds)
    np.testing.assert_allclose(float(scal[0]), float(loss), rtol=2e-6)
    assert rel_err(ds, gs) < 1e-5
    out = torch.empty(B, A, device=DEV)
    o.action_activation(tm.to(DEV), out, 1.0, True)
    np.testing.assert_allclose(out.cpu().numpy(), torch.tanh(tm).numpy(), rtol=2e-6, atol=2e-7)


# ------------------------------------------------------------------------------- clip + Adam
@pytest.mark.parametrize("n,n_clip,max_norm", [(558100, 558090, 0.5), (1000, 1000, 0.5), (300789, 300779, 0.0), (77, 70, 1e-3)])
def test_clip_adam_steps(n, n_clip, max_norm):
    o = ops()
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    pd = p0.clone().to(DEV)
    gd = torch.empty(n, device=DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    state = torch.zeros(4, dtype=torch.int32, device=DEV)
    skip = torch.zeros(1, device=DEV)
    gn = torch.zeros(1, device=DEV)
    ws = o.Workspace(torch.device(DEV))
    pr = p0.clone()
    adam = R.Adam([pr], 3e-3)
    for step in range(6):
        grad = torch.randn(n, generator=g) * (0.02 if step % 2 else 2.0)
        gd.copy_(grad)
        is_skip = step == 3
        skip.fill_(1.0 if is_skip else 0.0)
        o.clip_adam_step(pd, gd, m, v, n_clip, max_norm, 3e-3, 0.9, 0.999, 1e-8, state, skip, gn, ws)
        if not is_skip:
            gr = grad.clone()
            if max_norm > 0:
                total = R.clip_grad_norm([gr[:n_clip]], max_norm)
                np.testing.assert_allclose(float(gn), float(total), rtol=1e-5)
            adam.step([gr])
    assert int(state[0]) == 5
    np.testing.assert_allclose(pd.cpu().numpy(), pr.numpy(), rtol=0, atol=1e-6)
    assert rel_err(m, adam.m[0]) < 1e-5 and rel_err(v, adam.v[0]) < 1e-5


def test_one_call_forms_named_by_the_survey_boundary():
    """SURVEY.md 8(b) lists pm_adv_normalize_f32 and pm_mlp_fwd_f32 / pm_mlp_bwd_f32; the learner calls finer-grained forms
    (pm_moments_f64 + pm_normalize_apply_f32, pm_linear_*).  The listed names exist as entry points composed of those: the
    normalisation equals the reference expression `(x - x.mean()) / (x.std() + 1e-8)` (storage.py:114) and the two-call form
    bit for bit; the MLP forms equal the per-layer calls bit for bit and torch autograd (fp64) to fp32 round-off."""
    o = ops()
    g = torch.Generator().manual_seed(3)
    ws = o.Workspace(torch.device(DEV))
    x = (torch.randn(64 * 300, 1, generator=g) * 3 + 1).to(DEV)
    a = x.clone()
    o.adv_normalize(a, ws)
    b = x.clone()
    mom = torch.empty(2, dtype=torch.float64, device=DEV)
    o.moments(b, mom, ws)
    o.normalize_apply(b, mom, b.numel(), 1e-8)
    assert torch.equal(a, b)
    want = (x.cpu() - x.cpu().mean()) / (x.cpu().std() + 1e-8)
    np.testing.assert_allclose(a.cpu().numpy(), want.numpy(), rtol=2e-6, atol=2e-6)
    # the MLP of cfg 2's shape on a ragged batch
    M, dims = 1000, [53, 512, 512, 512, 10]
    xs = torch.randn(M, dims[0], generator=g).to(DEV)
    W = [(torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5).to(DEV) for i in range(4)]
    B_ = [(torch.randn(dims[i + 1], generator=g) * 0.1).to(DEV) for i in range(4)]
    hs = o.mlp_fwd(xs, W, B_, o.ACT_TANH)
    h, ref_h = xs, []
    for i in range(4):
        y = torch.empty(M, dims[i + 1], device=DEV)
        o.linear_fwd(h, W[i], B_[i], y, o.ACT_TANH if i < 3 else o.ACT_NONE)
        ref_h.append(y)
        h = y
    assert all(torch.equal(p, q) for p, q in zip(hs, ref_h))
    dy = torch.randn(M, dims[-1], generator=g).to(DEV)
    dws, dbs, dx = o.mlp_bwd(xs, W, hs, o.ACT_TANH, dy, ws, need_dx=True)
    xd = xs.double().requires_grad_(True)
    Wd = [w.double().requires_grad_(True) for w in W]
    Bd = [b.double().requires_grad_(True) for b in B_]
    hh = xd
    for i in range(4):
        hh = hh @ Wd[i].t() + Bd[i]
        if i < 3:
            hh = torch.tanh(hh)
    np.testing.assert_allclose(hs[-1].cpu().numpy(), hh.detach().cpu().numpy(), rtol=2e-5, atol=2e-5)
    (hh * dy.double()).sum().backward()
    for got, want_ in list(zip(dws, [w.grad for w in Wd])) + list(zip(dbs, [b.grad for b in Bd])) + [(dx, xd.grad)]:
        assert rel_err(got, want_.cpu()) < 2e-5


# ------------------------------------------------------------------------------- point-set ops
# wave-per-cloud variant: P <= 2048 with D == 3 (every points-per-lane instantiation, B not a multiple of 4,
# K > P); work-group variant: 2048 < P <= 8192 or D != 3; streaming variant beyond
@pytest.mark.parametrize("B,P,D,K", [(3, 1024, 3, 128), (2, 777, 3, 64), (2, 5000, 3, 256), (1, 20000, 3, 64), (2, 50, 4, 80),
                                     (5, 64, 3, 64), (6, 100, 3, 30), (3, 256, 3, 64), (2, 300, 3, 77), (7, 2048, 3, 96),
                                     (2, 40, 3, 50), (2, 1500, 4, 40)])
def test_fps_bit_exact(B, P, D, K):
    o = ops()
    g = torch.Generator().manual_seed(P)
    xyz = torch.rand(B, P, D, generator=g) * 2 - 1
    xyz[:, P // 3] = xyz[:, P // 5]                       # duplicated points -> exact ties
    idx = o.fps(xyz.to(DEV), K, o.Workspace(torch.device(DEV))).cpu().numpy()
    ref = R.fps(xyz.numpy(), K)
    assert np.array_equal(idx, ref.astype(np.int32))


@pytest.mark.parametrize("B,P,S,r,ns", [(2, 1024, 128, 0.3, 32), (1, 333, 50, 0.05, 16), (2, 1024, 64, 5.0, 64)])
def test_ball_query_and_group_bit_exact(B, P, S, r, ns):
    o = ops()
    g = torch.Generator().manual_seed(P + S)
    xyz = torch.rand(B, P, 3, generator=g) * 2 - 1
    ctr = xyz[:, torch.randperm(P, generator=g)[:S]].contiguous()
    if r < 0.1:
        ctr[:, 0] += 10.0                                   # an empty ball
    idx = o.ball_query(xyz.to(DEV), ctr.to(DEV), r, ns)
    ref = R.ball_query(xyz.numpy(), ctr.numpy(), r, ns)
    assert np.array_equal(idx.cpu().numpy(), ref)
    feat = torch.randn(B, P, 7, generator=g)
    out = o.group_points(feat.to(DEV), idx)
    assert np.array_equal(out.cpu().numpy(), R.group_points(feat.numpy(), ref))
    dout = torch.randn(B, S, ns, 7, generator=g)
    dfeat = o.group_points_bwd(dout.to(DEV), idx, P)
    dref = torch.zeros(B, P, 7, dtype=torch.float64)
    for b in range(B):
        dref[b].index_add_(0, torch.from_numpy(ref[b].reshape(-1).astype(np.int64)), dout[b].reshape(-1, 7).double())
    assert rel_err(dfeat, dref) < 1e-5
    # round 6: the gradient of the row gather runs in a fixed order (every source point's rows ascending: scatter_rows_det_kernel),
    # not through fp32 atomics -- bit-reproducible, every element written (a NaN-filled destination comes back finite), and equal
    # to the SEQUENTIAL fp32 sum in ascending row order; the same for the [xyz | feat | pad] rows of an unfused level (columns 3..)
    seq = torch.zeros(B, P, 7)
    for b in range(B):
        rows, d_ = ref[b].reshape(-1), dout[b].reshape(-1, 7)
        for r_ in range(rows.shape[0]):
            seq[b, rows[r_]] += d_[r_]
    assert torch.equal(dfeat.cpu(), seq)
    for _ in range(3):
        assert torch.equal(o.group_points_bwd(dout.to(DEV), idx, P), dfeat)
    ldo = 12
    drows = torch.randn(B * S * ns, ldo, generator=g)
    dc = o.group_concat_bwd(drows.to(DEV), idx, B, P, 7, ldo)
    seq2 = torch.zeros(B, P, 7)
    for b in range(B):
        rows = ref[b].reshape(-1)
        for r_ in range(rows.shape[0]):
            seq2[b, rows[r_]] += drows[b * S * ns + r_, 3:10]
    assert torch.equal(dc.cpu(), seq2) and torch.equal(o.group_concat_bwd(drows.to(DEV), idx, B, P, 7, ldo), dc)


def test_fast_tanh_accuracy():
    """pm_tanh (clamped odd rational P(x^2)/Q(x^2), one v_rcp + Newton step) against fp64 tanh on a dense grid, denormal
    and saturated inputs: a few ulp everywhere (torch's own tanh: 1 ulp), never beyond +-1."""
    o = ops()
    x = torch.cat([torch.linspace(-12, 12, 400001), torch.linspace(0.3, 0.4, 100001), torch.logspace(-30, -1, 5000),
                   torch.tensor([0.0, -0.0, 1e-38, 88.0, -88.0, 1e30, float("inf"), -float("inf")])]).float()
    out = torch.empty_like(x).to(DEV)
    o.action_activation(x.to(DEV), out, 1.0, True)
    ref = torch.tanh(x.double())
    got = out.cpu().double()
    err = (got - ref).abs()
    ulp = torch.maximum(ref.abs(), torch.tensor(1e-30, dtype=torch.float64)) * 2.0 ** -23
    assert float((err / ulp).max()) < 6.0, float((err / ulp).max())
    assert float(err.max()) < 3.0e-7
    assert abs(float(got[-2]) - 1.0) < 3e-7 and abs(float(got[-1]) + 1.0) < 3e-7 and got[-8] == 0.0   # +-inf, 0
    assert float(got.abs().max()) <= 1.0                              # never overshoots: 1 - h^2 stays >= 0


# ------------------------------------------------------------------------------- depth -> cloud (observation side)
def test_depth2pc_matches_reference_world_cloud_and_restated_sampling():
    """TSDFVolume.depth2pc (partmanip_amd/depth2tsdf.py): the back-projected, cropped world cloud is bit-identical
    to the REFERENCE's own output (fixture), the sampled indices to the CPU restatement's FPS."""
    from tests.golden import cases
    from partmanip_amd.depth2tsdf import TSDFVolume
    from partmanip_amd import ops
    c = cases.DEPTH2PC_CASES["depth2pc_small"]
    inp, fx = cases.depth2pc_inputs(c), load_fixture("depth2pc_small")
    vol = TSDFVolume(DEV, size=c["size"], resolution=10, _vol_origin=c["vol_origin"])
    vol.register_camera(inp["cam_pose"], np.asarray(c["intr"], dtype=np.float32), c["h"], c["w"], c["b"])
    depth = torch.from_numpy(inp["depth"]).to(DEV)
    lo = np.asarray(c["vol_origin"], dtype=np.float32)
    world = ops.depth_backproject(depth, vol.cam_pose, c["intr"][0][0], c["intr"][1][1], c["intr"][0][2], c["intr"][1][2],
                                  lo, np.float32(c["size"]) + lo)
    assert np.array_equal(world.cpu().numpy(), fx["world"])
    pc = vol.depth2pc(depth, K=c["K"])
    assert np.array_equal(pc.cpu().numpy(), fx["final_pc_1024"][:, :c["K"]])
    pc_full = vol.depth2pc(depth)                                   # the reference's K = 1024
    assert np.array_equal(pc_full.cpu().numpy(), fx["final_pc_1024"])


def test_depth2pc_streaming_fps_on_a_camera_sized_cloud():
    """One env x 3 views x 96 x 128 pixels = 36 864 points (> the register-resident limit): the streaming FPS
    variant against the restatement, indices bit-exact."""
    from partmanip_amd import ops
    g = torch.Generator().manual_seed(9)
    depth = (torch.rand(1, 3, 96, 128, generator=g) * 0.7 + 0.3)
    pose = torch.eye(4).repeat(3, 1, 1)
    pose[1, :3, 3] = torch.tensor([0.05, -0.02, 0.0])
    pose[2, :3, 3] = torch.tensor([-0.04, 0.03, 0.01])
    intr = [[120.0, 0.0, 63.5], [0.0, 110.0, 47.5], [0.0, 0.0, 1.0]]
    out_ref, world_ref, idx_ref = R.depth2pc(depth.numpy(), pose.numpy(), intr, 0.5, [-0.25, -0.25, 0.3], K=48, return_world=True)
    lo = np.asarray([-0.25, -0.25, 0.3], dtype=np.float32)
    world = ops.depth_backproject(depth.to(DEV), pose.to(DEV).contiguous(), 120.0, 110.0, 63.5, 47.5, lo, np.float32(0.5) + lo)
    assert np.array_equal(world.cpu().numpy(), world_ref)
    idx = ops.fps(world, 48, ops.Workspace(DEV))
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), idx_ref)


# (100, 120 000): two work-groups per cloud, chunks of up to 60 000 points -- beyond the 25 600 register + 10 176 LDS points of a
# work-group, i.e. the streamed region as well (the other shapes stay on chip since round 4)
@pytest.mark.parametrize("B,ld,K,pad", [(64, 140000, 40, False), (30, 100000, 33, True), (128, 70000, 24, False), (3, 260000, 20, True),
                                        (100, 120000, 16, False)])
def test_varlen_fps_on_several_work_groups_per_cloud_is_bit_exact(B, ld, K, pad):
    """Camera-sized variable-length clouds: G = 256 / B work-groups share a cloud, each keeps its chunk in registers / LDS (the
    rest streams) and the G candidates of a round are handed over inside the launch (pm_fps_varlen_f32 with the
    pm_fps_varlen_workspace_bytes workspace).  Indices bit-exact against the restatement for clouds of every kind in one batch:
    empty, register-sized (one work-group), chunks that fill registers only / registers + LDS / all three regions; exact ties
    (duplicated points, also ACROSS chunks: the lowest index must win on every work-group) and, with pad, K > length."""
    o = ops()
    g = np.random.default_rng(B + ld)
    pts = (g.random((B, ld, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    lengths = np.zeros(B, dtype=np.int32)
    kinds = [0, 17, 5000, 8192, 8193, 30000, ld // 2, ld - 3, ld]
    for b in range(B):
        lengths[b] = kinds[b % len(kinds)] if b < 2 * len(kinds) else int(g.integers(9000, ld + 1))
    for b in range(B):
        n = int(lengths[b])
        if n > 100:
            src = g.integers(0, n, size=n // 50)                       # 2 % duplicates of random earlier / later points
            dst = g.integers(0, n, size=n // 50)
            pts[b, dst] = pts[b, src]
            pts[b, n - 1] = pts[b, 1]                                   # a tie between the first and the last chunk
    ref = R.fps(pts, K, lengths)
    if not pad:                                                          # !pad keeps sampling (index 0 repeats) once a cloud is exhausted
        for b in range(B):
            n = int(lengths[b])
            if 0 < n < K:
                ref[b, n:] = 0
    idx = o.fps_varlen(torch.from_numpy(pts).to(DEV), torch.from_numpy(lengths).to(DEV), K, o.Workspace(torch.device(DEV)), pad=pad)
    got = idx.cpu().numpy().astype(np.int64)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, (bad[:5], got[bad[0][0], :8], ref[bad[0][0], :8], lengths[bad[0][0]])


def test_varlen_fps_hand_off_under_competing_load():
    """The same launch while another stream keeps the CUs busy with GEMMs of uneven length (work-groups of a cloud then start and
    finish their rounds at different times, some a round ahead of their partners -- the two alternating granule sets): the
    indices stay bit-identical over repeated calls and equal the restatement; no work-group gives up."""
    o = ops()
    B, ld, K = 40, 90000, 48
    g = np.random.default_rng(77)
    pts = (g.random((B, ld, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    lengths = g.integers(20000, ld + 1, size=B).astype(np.int32)
    ref = R.fps(pts, K, lengths)
    x, n = torch.from_numpy(pts).to(DEV), torch.from_numpy(lengths).to(DEV)
    ws = o.Workspace(torch.device(DEV))
    side = torch.cuda.Stream()
    mats = [torch.randn(m, m, device=DEV) for m in (512, 1536, 3072, 640)]
    torch.cuda.synchronize()
    for it in range(6):
        with torch.cuda.stream(side):
            for _ in range(12):
                for a in mats:
                    a @ a
        idx = o.fps_varlen(x, n, K, ws, pad=False)
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref), it
        assert not o.fps_varlen_gave_up(ws), it              # delayed hand-offs, never an exhausted poll budget
    torch.cuda.synchronize()


@pytest.mark.parametrize("pad", [False, True])
def test_varlen_fps_degrades_when_a_work_group_gives_up(pad, monkeypatch):
    """A multi-work-group launch whose partners cannot all be resident must DEGRADE, not fail (VERDICT r3 #8 / ADVICE): with the
    poll budget forced to zero every work-group that has to wait at all gives up in its first round, latches the flag, tells the
    others through the error word and returns; the launch queued behind re-samples the big clouds on one work-group each.  The
    indices are bit-exact all the same (empty / register-sized clouds of the batch included), no host round trip is involved,
    and the launch returns at once instead of spinning K times its budget."""
    import time
    o = ops()
    B, ld, K = 24, 40000, 64
    g = np.random.default_rng(5)
    pts = (g.random((B, ld, 3), dtype=np.float32) * 2 - 1).astype(np.float32)
    lengths = g.integers(9000, ld + 1, size=B).astype(np.int32)
    lengths[:3] = (0, 100, 8192)
    ref = R.fps(pts, K, lengths)
    if not pad:
        for b in range(B):
            if 0 < lengths[b] < K:
                ref[b, lengths[b]:] = 0
    x, n = torch.from_numpy(pts).to(DEV), torch.from_numpy(lengths).to(DEV)
    ws = o.Workspace(torch.device(DEV))
    assert int(o.lib.pm_fps_varlen_groups(B, ld, 3)) >= 2
    idx = o.fps_varlen(x, n, K, ws, pad=pad)                      # normal budget
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref) and not o.fps_varlen_gave_up(ws)
    torch.cuda.synchronize()                                      # (the poll budget is an ARGUMENT, pm_fps_config.spin_limit: no environment)
    t0 = time.perf_counter()
    idx = o.fps_varlen(x, n, K, ws, pad=pad, spin_limit=0)
    got = idx.cpu().numpy().astype(np.int64)
    dt = time.perf_counter() - t0
    assert o.fps_varlen_gave_up(ws), "the forced give-up did not happen"
    assert np.array_equal(got, ref)
    assert dt < 5.0, dt
    idx = o.fps_varlen(x, n, K, ws, pad=pad)                      # the next call clears the word and runs on several work-groups again
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref) and not o.fps_varlen_gave_up(ws)
    import ctypes
    cap = o.FPS_POLICY.struct(max_groups=1)                       # the cap: one work-group per cloud, no hand-offs at all
    assert int(o.lib.pm_fps_varlen_groups_cfg(B, ld, 3, ctypes.byref(cap))) == 1
    idx = o.fps_varlen(x, n, K, ws, pad=pad, max_groups=1)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref) and ws.fps_err is None
    # the caller says how many CUs a launch may occupy (a CU mask, a co-resident kernel): half of them -> half the work-groups per cloud
    full = int(o.lib.pm_fps_varlen_groups(B, ld, 3))
    half = o.FPS_POLICY.struct(resident_cus=128)
    assert int(o.lib.pm_fps_varlen_groups_cfg(B, ld, 3, ctypes.byref(half))) == max(1, min(full, 128 // B))
    idx = o.fps_varlen(x, n, K, ws, pad=pad, resident_cus=128)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref) and not o.fps_varlen_gave_up(ws)
    assert not o.FPS_POLICY.gave_up_cap                           # the forced give-up above reported, it did not cap the process


def test_tsdf_integrate_matches_reference():
    """TSDFVolume.integrate (pm_tsdf_integrate_f32) against the REFERENCE's own volume (fixture), bit for bit."""
    from partmanip_amd.depth2tsdf import TSDFVolume
    c = cases.DEPTH2PC_CASES["depth2pc_small"]
    inp, fx = cases.depth2pc_inputs(c), load_fixture("depth2pc_small")
    vol = TSDFVolume(DEV, size=c["size"], resolution=10, _vol_origin=c["vol_origin"])
    vol.register_camera(inp["cam_pose"], np.asarray(c["intr"], dtype=np.float32), c["h"], c["w"], c["b"])
    depth = torch.from_numpy(inp["depth"]).to(DEV)
    out = vol.integrate(depth)
    assert tuple(out.shape) == (c["b"], 10, 10, 10)
    # own registration tables (host bmm: last bit of the voxel depths is CPU-dependent) -> fp32 round-off class
    assert np.array_equal(vol._pix_idx.cpu().numpy(), fx["tsdf_pix_idx"])
    np.testing.assert_allclose(out.cpu().numpy(), fx["tsdf"], rtol=0, atol=2e-6)
    # the kernel itself, on the reference's tables: bit-identical
    vol._pix_idx = torch.from_numpy(fx["tsdf_pix_idx"]).to(DEV)
    vol.pix_z = torch.from_numpy(fx["tsdf_pix_z"]).to(DEV)
    assert np.array_equal(vol.integrate(depth).cpu().numpy(), fx["tsdf"])


def test_sparse_voxel_matches_reference():
    """TSDFVolume.sparse_voxel (select -> pm_fps_varlen_f32 with padding -> gather) against the REFERENCE's own
    (b, 1024, 4) output (fixture), bit for bit, on the reference's registration tables."""
    from partmanip_amd.depth2tsdf import TSDFVolume
    c = cases.DEPTH2PC_CASES["depth2pc_small"]
    inp, fx = cases.depth2pc_inputs(c), load_fixture("depth2pc_small")
    vol = TSDFVolume(DEV, size=c["size"], resolution=10, _vol_origin=c["vol_origin"])
    vol.register_camera(inp["cam_pose"], np.asarray(c["intr"], dtype=np.float32), c["h"], c["w"], c["b"])
    vol._pix_idx = torch.from_numpy(fx["tsdf_pix_idx"]).to(DEV)
    vol.pix_z = torch.from_numpy(fx["tsdf_pix_z"]).to(DEV)
    out = vol.sparse_voxel(torch.from_numpy(inp["depth"]).to(DEV))
    assert tuple(out.shape) == (c["b"], 1024, 4) and out.dtype == torch.float32
    assert np.array_equal(out.cpu().numpy(), fx["sparse_voxel"])


@pytest.mark.parametrize("res,band,K", [(12, 0.9, 256), (20, 0.5, 1024), (50, 0.25, 1024), (9, 0.0, 64), (16, 2.0, 128)])
def test_sparse_voxel_kernels_against_restatement(res, band, K):
    """Random volumes with many more band voxels than K (the 50^3 production grid: ~31k candidates -> the streaming
    FPS), none at all (band 0), and every voxel (band 2): selection order, sampling and gather vs the restatement."""
    o = ops()
    g = torch.Generator().manual_seed(res * 7 + K)
    vol = torch.rand(3, res, res, res, generator=g) * 2 - 1
    vol[1, :, : res // 2] = 1.0                                      # ragged: env 1 has about half the candidates
    if band == 0.0:
        ref = np.zeros((3, K, 4), np.float32)
        ref[..., 3] = vol[:, 0, 0, 0].numpy()[:, None]
    else:
        ref = R.tsdf_sparse_voxel(vol.numpy(), K=K, lo=-band, hi=band).numpy()
    out = o.tsdf_sparse_voxel(vol.to(DEV), K, -band, band, o.Workspace(DEV))
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("P,frac_valid,K", [(5000, 0.3, 64), (40000, 0.1, 48), (3000, 0.0, 16), (2000, 1.0, 32), (700, 0.01, 40)])
def test_depth_compact_then_varlen_fps_selects_the_same_points_as_the_full_cloud(P, frac_valid, K):
    """Crop compaction + variable-length FPS (what TSDFVolume.depth2pc runs) against FPS over the FULL cloud by the
    CPU restatement: the selected POINTS are identical, including when fewer distinct points than K exist."""
    o = ops()
    g = torch.Generator().manual_seed(P)
    B = 3
    xyz = torch.rand(B, P, 3, generator=g) * 2 - 1
    keep = torch.rand(B, P, 1, generator=g) < frac_valid
    if frac_valid > 0:
        keep[1, 0] = True                                  # env 1 starts with a valid point, env 0/2 as drawn
        keep[0, 0] = False
    world = (xyz * keep).contiguous()
    compact, lengths = o.depth_compact(world.to(DEV))
    n = lengths.cpu().numpy()
    nz = (world != 0).any(-1)
    want_n = nz.sum(1).numpy() + (~nz).any(1).numpy().astype(np.int64)
    assert np.array_equal(n, want_n)
    idx = o.fps_varlen(compact, lengths, K, o.Workspace(torch.device(DEV)))
    got = o.group_points(compact, idx.view(B, K, 1)).view(B, K, 3).cpu().numpy()
    ref_idx = R.fps(world.numpy(), K)
    want = np.take_along_axis(world.numpy(), ref_idx[..., None].repeat(3, axis=-1), axis=1)
    assert np.array_equal(got, want)


# ------------------------------------------------------------------------------- fused set-abstraction level
@pytest.mark.parametrize("dims,P,S,cf", [((64, 64, 128), 1024, 256, 0), ((128, 128, 256), 256, 64, 128)])
def test_sa_kernels_are_invariant_to_neighbour_order_at_bench_size(dims, P, S, cf):
    """Size-independent property at the benchmark's per-level shapes (512 clouds): max-pooling over a group does not
    depend on the order of its 32 neighbours.  Forward: pooled features bit-identical, arg-max follows the
    permutation.  Backward (saved layer 2 and recompute): weight gradients equal to fp32 summation round-off, the
    scattered dY likewise."""
    o = ops()
    B, (C1, C2, C3) = 512, dims
    g = torch.Generator().manual_seed(P + cf)
    xyz = (torch.rand(B, P, 3, generator=g) * 2 - 1).to(DEV)
    ws = o.Workspace(torch.device(DEV))
    idx_c = o.fps(xyz, S, ws)
    centers = o.group_points(xyz, idx_c.view(B, S, 1)).view(B, S, 3)
    idx = o.ball_query(xyz, centers, 0.3 if cf == 0 else 0.6, 32)
    perm = torch.stack([torch.randperm(32, generator=g) for _ in range(B * S)]).view(B, S, 32).to(DEV)
    idx_p = torch.gather(idx, 2, perm).contiguous()
    ldw1 = (3 + cf + 3) // 4 * 4
    W1 = (torch.randn(C1, ldw1, generator=g) * 0.3).to(DEV)
    W2 = (torch.randn(C2, C1, generator=g) / C1 ** 0.5).to(DEV)
    W3 = (torch.randn(C3, C2, generator=g) / C2 ** 0.5).to(DEV)
    b1, b2, b3 = ((torch.randn(c, generator=g) * 0.1).to(DEV) for c in dims)
    packed = torch.empty(int(o.lib.pm_sa_packed_elems(*dims)), device=DEV)
    o.sa_pack(W2, W3, packed)
    Y = None
    if cf:
        feat = (torch.randn(B * P, cf, generator=g) * 0.5).to(DEV)
        Y = torch.empty(B * P, C1, device=DEV)
        o.linear_fwd(feat, W1[:, 3:3 + cf], None, Y, o.ACT_NONE)
    dpooled = torch.randn(B * S, C3, generator=g).to(DEV)
    res = []
    for ii, save in ((idx, True), (idx_p, True), (idx, False)):
        pooled = torch.empty(B * S, C3, device=DEV)
        h2 = torch.empty(B * S * 32, C2, device=DEV) if save else None
        arg = o.sa_fwd(xyz, centers, ii, Y, W1, b1, b2, b3, packed, dims, pooled, h2)
        grads = [torch.empty_like(t_) for t_ in (W1, b1, W2, b2, W3, b3)]
        dY = torch.zeros(B * P, C1, device=DEV) if cf else None
        o.sa_bwd(xyz, centers, ii, Y, W1, b1, b2, W3, packed, dims, pooled, arg, dpooled, *grads, dY, ws, h2)
        res.append((pooled, arg, grads, dY))
    (p0, a0, g0, y0), (p1, a1, g1, y1), (p2, a2, g2, y2) = res
    assert torch.equal(p0, p1) and torch.equal(p0, p2) and torch.equal(a0, a2)
    # the arg-max row of the permuted run points at the same neighbour wherever the maximum is unique
    src0 = torch.gather(idx.view(B * S, 32), 1, a0.long())
    src1 = torch.gather(idx_p.view(B * S, 32), 1, a1.long())
    flips = float((src0 != src1).float().mean())               # exact fp32 ties between two distinct neighbours
    assert flips < 1e-5, flips                                 # (about one in 10^7 (group, channel) pairs)
    for k, (x0, x1, x2) in enumerate(zip(g0, g1, g2)):
        gw = x0[:, :3] if k == 0 else x0                       # dW1: only the xyz columns are written by the kernel
        hw = x1[:, :3] if k == 0 else x1
        rw = x2[:, :3] if k == 0 else x2
        scale = float(gw.abs().max()) + 1e-12
        # permuted neighbours = another summation order over millions of fp32 rows (round-off of long, cancelling
        # sums), plus an O(1) change in ONE row of dW3 for every exact-tie flip above
        d = (gw - hw).abs()
        assert float((d > 2e-3 * scale).float().mean()) < 1e-3 and float(d.max()) < 2e-2 * scale, k
        # saved vs recomputed layer 2, same order: the two backward paths see bit-identical H2
        assert float((gw - rw).abs().max()) < 1e-6 * scale, k
    if cf:
        dyd = (y0 - y1).abs()                                  # a tie flip moves one row's gradient to another source point
        assert float((dyd > 1e-4 * float(y0.abs().max())).float().mean()) < 1e-4
        assert float((y0 - y2).abs().max()) < 1e-5 * float(y0.abs().max())      # global fp32 atomics: order varies


@pytest.mark.parametrize("dims,P,S,cf,radius", [((64, 64, 128), 1024, 256, 0, 0.2), ((128, 128, 256), 256, 64, 128, 0.4),
                                                 ((64, 64, 128), 300, 40, 0, 5.0), ((128, 128, 256), 200, 24, 128, 0.05),
                                                 ((128, 128, 256), 96, 48, 128, 5.0)])
def test_sa_packed_rows_equal_the_dense_level(dims, P, S, cf, radius):
    """The duplicate-free form of the fused level (pm_sa_plan_i32 + pm_sa_*_packed_f32) against the dense kernels on the
    same neighbourhood table: the plan's tables against a numpy restatement (distinct rows per group = entries that
    differ from entry 0, whole groups per tile, tile limits respected), pooled / arg / the saved layer 2 BIT-identical,
    every gradient equal to fp32 summation order.  Radii: the benchmark's (3.9 / 5.4 distinct rows of 32), 5.0 (every
    ball full: 32 distinct rows, the packed form degenerates to the dense one) and 0.05 (most balls hold the centre only)."""
    o = ops()
    B, (C1, C2, C3) = 37, dims
    g = torch.Generator().manual_seed(P + cf + int(radius * 100))
    xyz = (torch.rand(B, P, 3, generator=g) * 2 - 1).to(DEV)
    ws = o.Workspace(torch.device(DEV))
    idx_c = o.fps(xyz, S, ws)
    centers = o.group_points(xyz, idx_c.view(B, S, 1)).view(B, S, 3).contiguous()
    if radius < 0.1:
        centers[:, 0] += 10.0                                  # empty balls: the row is all zeros (point 0, once)
    idx = o.ball_query(xyz, centers, radius, 32)
    plan = o.sa_plan(idx, xyz, centers, dims, ws)
    R_, T_ = plan.counts()
    # ---- the plan against numpy
    ii = idx.cpu().numpy().reshape(B * S, 32)
    cnt = 1 + (ii[:, 1:] != ii[:, :1]).sum(1)
    grow = np.concatenate([[0], np.cumsum(cnt)])
    assert R_ == grow[-1] and np.array_equal(plan.grow.cpu().numpy(), grow)
    rm = plan.rowmap.cpu().numpy()[:R_]
    want_g = np.repeat(np.arange(B * S), cnt)
    want_sp = np.concatenate([np.concatenate([r[:1], r[1:][r[1:] != r[0]]]) for r in ii]) + (want_g // S) * P
    assert np.array_equal(rm[:, 0], want_sp)
    tr, tg = o.sa_packed_tile(dims)
    tl = plan.tiles.cpu().numpy()[:T_]
    # rowmap.y = (group - first group of its tile) << 8 | row inside the group; relxyz = xyz[source] - centre[group] (fp32 sub)
    tile_of_row = np.repeat(np.arange(T_), tl[:, 3])
    assert np.array_equal(rm[:, 1] >> 8, want_g - tl[tile_of_row, 1]) and np.array_equal(rm[:, 1] & 255, np.arange(R_) - grow[want_g])
    rel = plan.relxyz.cpu().numpy()[:R_]
    want_rel = xyz.cpu().numpy().reshape(-1, 3)[want_sp] - centers.cpu().numpy().reshape(-1, 3)[want_g]
    assert np.array_equal(rel[:, :3], want_rel) and not rel[:, 3].any()
    assert tl[0, 0] == 0 and tl[0, 1] == 0 and np.array_equal(tl[1:, 0], tl[:-1, 0] + tl[:-1, 3])
    assert np.array_equal(tl[1:, 1], tl[:-1, 1] + tl[:-1, 2]) and tl[-1, 1] + tl[-1, 2] == B * S and tl[-1, 0] + tl[-1, 3] == R_
    assert tl[:, 3].max() <= tr and tl[:, 2].max() <= tg and tl[:, 2].min() >= 1
    assert np.array_equal(tl[:, 3], grow[tl[:, 1] + tl[:, 2]] - grow[tl[:, 1]])          # whole groups
    assert np.all(tl[:, 1] // S == (tl[:, 1] + tl[:, 2] - 1) // S)                         # never across clouds
    # ---- kernels
    ldw1 = (3 + cf + 3) // 4 * 4
    W1 = (torch.randn(C1, ldw1, generator=g) * 0.3).to(DEV)
    W2 = (torch.randn(C2, C1, generator=g) / C1 ** 0.5).to(DEV)
    W3 = (torch.randn(C3, C2, generator=g) / C2 ** 0.5).to(DEV)
    b1, b2, b3 = ((torch.randn(c, generator=g) * 0.1).to(DEV) for c in dims)
    packed = torch.empty(int(o.lib.pm_sa_packed_elems(*dims)), device=DEV)
    o.sa_pack(W2, W3, packed)
    Y = None
    if cf:
        feat = (torch.randn(B * P, cf, generator=g) * 0.5).to(DEV)
        Y = torch.empty(B * P, C1, device=DEV)
        o.linear_fwd(feat, W1[:, 3:3 + cf], None, Y, o.ACT_NONE)
    dpooled = torch.randn(B * S, C3, generator=g).to(DEV)
    out = {}
    for mode in ("dense", "packed", "packed_recompute"):
        pooled = torch.full((B * S, C3), float("nan"), device=DEV)
        h2 = torch.zeros(B * S * 32, C2, device=DEV) if mode != "packed_recompute" else None
        grads = [torch.full_like(t_, float("nan")) for t_ in (W1, b1, W2, b2, W3, b3)]
        dY = torch.zeros(B * P, C1, device=DEV) if cf else None
        if mode == "dense":
            arg = o.sa_fwd(xyz, centers, idx, Y, W1, b1, b2, b3, packed, dims, pooled, h2)
            o.sa_bwd(xyz, centers, idx, Y, W1, b1, b2, W3, packed, dims, pooled, arg, dpooled, *grads, dY, ws, h2)
        else:
            arg = o.sa_fwd_packed(plan, Y, W1, b1, b2, b3, packed, dims, pooled, h2)
            o.sa_bwd_packed(plan, Y, W1, b1, b2, W3, packed, dims, pooled, arg, dpooled, *grads, dY, ws, h2)
        out[mode] = (pooled, arg, h2, grads, dY)
    pd, ad, hd, gd, yd = out["dense"]
    for mode in ("packed", "packed_recompute"):
        pp, ap, hp, gp, yp = out[mode]
        assert torch.equal(pd, pp) and torch.equal(ad, ap), mode
        if hp is not None:                                     # packed row r of group g = dense row g*32 + (r - grow[g])
            dense_rows = torch.from_numpy(want_g * 32 + (np.arange(R_) - grow[want_g])).to(DEV)
            assert torch.equal(hd[dense_rows], hp[:R_])
        for k_, (x0, x1) in enumerate(zip(gd, gp)):
            a_ = x0[:, :3] if k_ == 0 else x0                  # dW1: only the xyz columns are written by the kernels
            b_ = x1[:, :3] if k_ == 0 else x1
            scale = float(a_.abs().max()) + 1e-12
            assert bool(torch.isfinite(b_).all()) and float((a_ - b_).abs().max()) < 2e-5 * scale, (mode, k_)
        if cf:
            assert float((yd - yp).abs().max()) < 1e-5 * float(yd.abs().max())
    if not cf:
        return
    # ---- the deterministic gradient of the per-source-point layer-1 rows: no fp32 atomics (pm_sa_plan_inverse_i32 + dz1 rows +
    # pm_sa_dy_segsum_f32).  The inverse table against numpy (every point's packed rows, ascending), the sums against a float64
    # scatter-add of the same dz1 rows, and five repetitions of the whole backward bit-identical.
    plan_i = o.sa_plan(idx, xyz, centers, dims, ws, inverse=True)
    start, rows = plan_i.inv_start.cpu().numpy(), plan_i.inv_rows.cpu().numpy()[:R_]
    order = np.argsort(want_sp, kind="stable")                 # stable: ascending packed row inside a point
    assert np.array_equal(start, np.concatenate([[0], np.cumsum(np.bincount(want_sp, minlength=B * P))]))
    assert np.array_equal(rows, order)
    runs = []
    for rep in range(5):
        pooled = torch.empty(B * S, C3, device=DEV)
        h2 = torch.empty(B * S * 32, C2, device=DEV)
        grads = [torch.full_like(t_, float("nan")) for t_ in (W1, b1, W2, b2, W3, b3)]
        dz1 = torch.full((R_ + 3, C1), float("nan"), device=DEV)[:R_]
        dYd = torch.full((B * P, C1), float("nan"), device=DEV)
        arg = o.sa_fwd_packed(plan_i, Y, W1, b1, b2, b3, packed, dims, pooled, h2)
        o.sa_bwd_packed(plan_i, Y, W1, b1, b2, W3, packed, dims, pooled, arg, dpooled, *grads, None, ws, h2, dz1=dz1)
        o.sa_dy_segsum(plan_i, dz1, dYd)
        runs.append((dYd, dz1, grads))
    dYd, dz1, grads = runs[0]
    assert bool(torch.isfinite(dYd).all()) and bool(torch.isfinite(dz1).all())
    want = torch.zeros(B * P, C1, dtype=torch.float64, device=DEV).index_add_(0, torch.from_numpy(want_sp).to(DEV), dz1.double())
    assert float((dYd.double() - want).abs().max()) < 1e-6 * float(want.abs().max())
    assert float((dYd - out["packed"][4]).abs().max()) < 1e-5 * float(dYd.abs().max())          # the atomic path's sums
    for k_, (x0, x1) in enumerate(zip(out["packed"][3], grads)):                                   # the other gradients: untouched
        assert torch.equal(x0[:, :3] if k_ == 0 else x0, x1[:, :3] if k_ == 0 else x1), k_
    for dYr, dzr, gr in runs[1:]:
        assert torch.equal(dYr, dYd) and torch.equal(dzr, dz1) and all(
            torch.equal(a_[:, :3] if k_ == 0 else a_, b_[:, :3] if k_ == 0 else b_) for k_, (a_, b_) in enumerate(zip(gr, grads)))
    # ---- the sums consumed where they are formed (pm_sa_dy_consume_f32): its optional dY copy BIT-identical to the segmented-sum pass,
    # dfeat = dY W1f and dW1[:, 3:3+cf] = dY^T feat against float64 products of that dY, pad columns zeroed, xyz columns untouched,
    # three repetitions bit-identical (npoints = 37 x 200 is not a multiple of the 64-point tile: the ragged last tile)
    assert o.sa_dy_consume_supported(C1, cf)
    pw = torch.empty(int(o.lib.pm_sa_dy_consume_packed_elems(C1, cf)), device=DEV)
    o.sa_dy_consume_pack(W1, cf, pw)
    cons = []
    for rep in range(3):
        dfeat = torch.full((B * P, cf), float("nan"), device=DEV)
        dW1c = torch.full_like(W1, float("nan"))
        dW1c[:, :3] = 7.0
        dYc = torch.full((B * P, C1), float("nan"), device=DEV)
        o.sa_dy_consume(plan_i, dz1, feat, pw, dfeat, dW1c, ws, dY=dYc if rep == 0 else None)
        cons.append((dfeat, dW1c))
        if rep == 0:
            assert torch.equal(dYc, dYd)
    dfeat, dW1c = cons[0]
    want_df = dYd.double() @ W1[:, 3:3 + cf].double()
    want_dw = dYd.double().t() @ feat.double()
    assert float((dfeat.double() - want_df).abs().max()) < 2e-6 * float(want_df.abs().max())
    assert float((dW1c[:, 3:3 + cf].double() - want_dw).abs().max()) < 5e-6 * float(want_dw.abs().max())
    assert bool((dW1c[:, :3] == 7.0).all()) and not bool(dW1c[:, 3 + cf:].any())
    for dfr, dwr in cons[1:]:
        assert torch.equal(dfr, dfeat) and torch.equal(dwr, dW1c)


def test_grouped_linear_ops_equal_the_single_problem_ops():
    """pm_linear_*_group_f32: several problems per launch, each bit-identical to its own single-problem launch; the
    split-K slabs of the grouped weight gradient add up (in slab order) to the single call's result to fp32 round-off."""
    from partmanip_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    ws = ops.Workspace(DEV)
    probs = [(2048, 512, 512), (2048, 512, 256), (1024, 53, 512), (2048, 512, 10)]
    xs = [r(M, K) for M, K, N in probs]
    wts = [r(N, K) / K ** 0.5 for M, K, N in probs]
    bs = [r(N) for M, K, N in probs]
    y1 = [torch.empty(M, N, device=DEV) for M, K, N in probs]
    y2 = [torch.empty(M, N, device=DEV) for M, K, N in probs]
    for x, w, b, y in zip(xs, wts, bs, y1):
        ops.linear_fwd(x, w, b, y, ops.ACT_TANH)
    ops.linear_fwd_group([(x, w, b, y, ops.ACT_TANH) for x, w, b, y in zip(xs, wts, bs, y2)][:2])
    ops.linear_fwd_group([(x, w, b, y, ops.ACT_TANH) for x, w, b, y in zip(xs, wts, bs, y2)][2:])
    for (M, K, N), a, b_ in zip(probs, y1, y2):              # <= 16 outputs: the single call runs the skinny VALU kernel
        assert torch.equal(a, b_) if N > 16 else float((a - b_).abs().max()) < 2e-5
    dys = [r(M, N) for M, K, N in probs]
    dx1 = [torch.empty(M, K, device=DEV) for M, K, N in probs]
    dx2 = [torch.empty(M, K, device=DEV) for M, K, N in probs]
    for dy, w, x, dx in zip(dys, wts, xs, dx1):
        ops.linear_bwd_data(dy, w, x, dx, ops.ACT_TANH)
    ops.linear_bwd_data_group([(dy, w, x, dx, ops.ACT_TANH) for dy, w, x, dx in zip(dys, wts, xs, dx2)])
    for (M, K, N), a, b_ in zip(probs, dx1, dx2):
        assert torch.equal(a, b_) if N > 16 else float((a - b_).abs().max()) < 2e-5 * max(1.0, float(a.abs().max()))
    # weight gradients: 4 slabs per problem, summed in slab order
    S = 4
    for (M, K, N), dy, x in zip(probs, dys, xs):
        dw1, db1 = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        ops.linear_bwd_weight(dy, x, dw1, db1, ws)
        stride = (N * K + N + 3) // 4 * 4
        slabs = torch.zeros(S * stride, device=DEV)
        ops.linear_bwd_weight_group([(dy, x, slabs[:N * K].view(N, K), slabs[N * K:N * K + N], stride)], S)
        tot = slabs.view(S, stride).sum(0)
        ref = (dy.double().t() @ x.double())
        assert float((tot[:N * K].view(N, K).double() - ref).abs().max() / ref.abs().max()) < 2e-6
        assert float((dw1.double() - ref).abs().max() / ref.abs().max()) < 2e-6
        np.testing.assert_allclose(tot[N * K:N * K + N].cpu().numpy(), dy.double().sum(0).float().cpu().numpy(), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(db1.cpu().numpy(), dy.double().sum(0).float().cpu().numpy(), rtol=1e-5, atol=1e-4)


def test_grouped_clip_adam_equals_two_single_steps():
    from partmanip_amd import ops
    g = torch.Generator(device=DEV).manual_seed(6)
    items1, items2, ref = [], [], []
    for n, n_clip, S in ((10007, 9000, 3), (5000, 5000, 0)):
        p = torch.randn(n, device=DEV, generator=g)
        gr = torch.randn(n, device=DEV, generator=g)
        extra = torch.randn(max(S, 1) * n, device=DEV, generator=g)
        gsum = gr.clone()
        for z in range(S):
            gsum[:n_clip] += extra[z * n:z * n + n_clip]
        a = dict(p=p.clone(), g=gsum.clone(), m=torch.zeros(n, device=DEV), v=torch.zeros(n, device=DEV),
                 state=torch.zeros(4, dtype=torch.int32, device=DEV), gnorm=torch.zeros(1, device=DEV), ws=ops.Workspace(DEV))
        ops.clip_adam_step(a["p"], a["g"], a["m"], a["v"], n_clip, 0.5, 1e-3, 0.9, 0.999, 1e-8, a["state"], None, a["gnorm"], a["ws"])
        ref.append(a)
        b = dict(p=p.clone(), g=gr.clone(), m=torch.zeros(n, device=DEV), v=torch.zeros(n, device=DEV),
                 state=torch.zeros(4, dtype=torch.int32, device=DEV), gnorm=torch.zeros(1, device=DEV), ws=ops.Workspace(DEV),
                 n_clip=n_clip, max_norm=0.5, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, skip_flag=None, extra=extra, extra_stride=n,
                 n_sum=n_clip, n_extra=S)
        items2.append(b)
    ops.clip_adam_group(items2)
    for a, b in zip(ref, items2):
        assert torch.equal(a["g"], b["g"]) and torch.equal(a["p"], b["p"]) and torch.equal(a["m"], b["m"])
        assert torch.equal(a["v"], b["v"]) and int(b["state"][0]) == 1 and torch.equal(a["gnorm"], b["gnorm"])


@pytest.mark.parametrize("M,K,N,act", [(2048, 512, 10, 0), (2048, 512, 1, 0), (2048, 32, 10, 1), (300, 128, 16, 1), (4096, 1028, 7, 0)])
def test_skinny_linear_layers(M, K, N, act):
    """Layers with <= 16 outputs (policy / value heads) run as row-wise VALU kernels instead of 64-wide GEMM tiles: forward,
    data gradient (with the activation derivative of the layer input) and weight / bias gradient against fp64."""
    from partmanip_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + K + N)
    x = torch.tanh(torch.randn(M, K, device=DEV, generator=g))
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    b = torch.randn(N, device=DEV, generator=g)
    y = torch.empty(M, N, device=DEV)
    ops.linear_fwd(x, w, b, y, ops.ACT_TANH if act else ops.ACT_NONE)
    ref = x.double() @ w.double().t() + b.double()
    ref = torch.tanh(ref) if act else ref
    assert float((y.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    dy = torch.randn(M, N, device=DEV, generator=g)
    dx = torch.empty(M, K, device=DEV)
    ops.linear_bwd_data(dy, w, x, dx, ops.ACT_TANH)
    rdx = (dy.double() @ w.double()) * (1 - x.double() ** 2)
    assert float((dx.double() - rdx).abs().max()) < 2e-5 * max(1.0, float(rdx.abs().max()))
    dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    ops.linear_bwd_weight(dy, x, dw, db, ops.Workspace(DEV))
    rdw, rdb = dy.double().t() @ x.double(), dy.double().sum(0)
    assert float((dw.double() - rdw).abs().max()) < 3e-5 * max(1.0, float(rdw.abs().max()))
    assert float((db.double() - rdb).abs().max()) < 3e-5 * max(1.0, float(rdb.abs().max()))


@pytest.mark.parametrize("G,ns,C", [(5, 4096, 32), (3, 300, 64), (2, 257, 256), (7, 1024, 16), (4, 32, 32), (3, 700, 24)])
def test_maxpool_rows_value_and_lowest_arg(G, ns, C):
    """Pooling over a group's rows (long groups: one work-group per group, chunked scan + LDS merge; short ones / channel counts
    that do not divide 256: the serial scan): the greatest value and the LOWEST row attaining it (values quantised -> many ties)."""
    g = torch.Generator().manual_seed(G * 31 + ns)
    x = (torch.randn(G * ns, C, generator=g) * 4).round() / 4
    out = torch.empty(G, C + 3, device=DEV)[:, :C]
    arg = ops().maxpool_rows(x.to(DEV), G, ns, out)
    xr = x.view(G, ns, C).numpy()
    np.testing.assert_array_equal(out.cpu().numpy(), xr.max(1))
    np.testing.assert_array_equal(arg.cpu().numpy(), xr.argmax(1).astype(np.int32))


@pytest.mark.parametrize("rows,J,C,N", [(1000, 27, 32, 32), (333, 8, 32, 64), (4097, 27, 64, 64), (700, 32, 4, 32), (129, 27, 32, 128),
                                        (50001, 27, 64, 32), (3000, 32, 32, 16), (2049, 8, 64, 32),
                                        # >= 512 big tiles and <= 32 output columns: the forward's 128 x 32 tile (4 x 1 waves)
                                        (70001, 32, 4, 32), (66000, 3, 32, 32), (65600, 27, 32, 32)])
def test_sparse_conv_entry_points_against_torch(rows, J, C, N):
    """The three gathered-GEMM entry points (the (rows x J*C) operand exists only inside the LDS-DMA loader) against a plain
    torch fp32 gather + matmul of the same op: random neighbour tables with ~30 % absent neighbours (-1 -> zero rows),
    row counts that do not fill the 128 / 64-row tiles."""
    o = ops()
    g = torch.Generator().manual_seed(rows + J)
    nsrc = rows + 17
    src = torch.randn(nsrc, C, generator=g)
    idx = torch.randint(0, nsrc, (rows, J), generator=g, dtype=torch.int32)
    idx[torch.rand(rows, J, generator=g) < 0.3] = -1
    w = torch.randn(N, J * C, generator=g) / (J * C) ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    zero = torch.zeros(256, device=DEV)
    srcp = torch.cat([src, torch.zeros(1, C)])                       # index -1 -> the appended zero row
    cols = srcp[idx.long()].reshape(rows, J * C)
    # forward (+ bias + tanh)
    y = torch.empty(rows, N, device=DEV)
    o.sparse_conv_fwd(src.to(DEV), idx.to(DEV), C, w.to(DEV), b.to(DEV), y, o.ACT_TANH, zero)
    # (reference in fp64: the fp32 round-off of a K = J*C reduction, ~1e-6 at K = 1728, is the only difference)
    np.testing.assert_allclose(y.cpu().numpy(), torch.tanh(cols.double() @ w.double().t() + b.double()).numpy(), rtol=2e-5, atol=1e-5)
    # weight gradient
    dy = torch.randn(rows, N, generator=g)
    dw, db = torch.empty(N, J * C, device=DEV), torch.empty(N, device=DEV)
    o.sparse_conv_bwd_weight(dy.to(DEV), src.to(DEV), idx.to(DEV), C, dw, db, zero, o.Workspace(DEV))
    ref = dy.double().t() @ cols.double()
    assert rel_err(dw, ref) < 2e-5 and rel_err(db, dy.double().sum(0)) < 2e-5
    # data-gradient form: dx = (gather(dy, idx_t) @ wt.T) * (1 - h^2), dy rows N wide, output C wide
    idx_t = torch.randint(0, rows, (rows, J), generator=g, dtype=torch.int32)
    idx_t[torch.rand(rows, J, generator=g) < 0.3] = -1
    idx_t[5] = -1                                                     # a row that is nobody's neighbour: exactly zero
    wt = torch.randn(C, J * N, generator=g) / (J * N) ** 0.5
    h = torch.tanh(torch.randn(rows, C, generator=g))
    dx = torch.empty(rows, C, device=DEV)
    o.sparse_conv_bwd_data(dy.to(DEV), idx_t.to(DEV), wt.to(DEV), h.to(DEV), dx, o.ACT_TANH, zero)
    dyp = torch.cat([dy, torch.zeros(1, N)])
    refx = (dyp[idx_t.long()].reshape(rows, J * N).double() @ wt.double().t()) * (1 - h.double() ** 2)
    assert rel_err(dx, refx) < 2e-5
    assert float(dx[5].abs().max()) == 0.0


@pytest.mark.parametrize("rows,J,C,N", [(700, 28, 16, 32), (129, 8, 32, 64), (2050, 27, 32, 32)])
def test_sparse_conv_bwd_data_scatter_against_torch(rows, J, C, N):
    """Data gradient by scatter (non-overlapping patches: every destination row has at most one (r, j)): against a torch
    matmul + index_put; destination rows nobody maps to keep their contents, absent taps (-1) are dropped."""
    o = ops()
    g = torch.Generator().manual_seed(rows * 7 + J)
    ndst = rows * J + 11
    perm = torch.randperm(ndst, generator=g)[: rows * J].reshape(rows, J).to(torch.int32)
    perm[torch.rand(rows, J, generator=g) < 0.2] = -1
    dy = torch.randn(rows, N, generator=g)
    w = torch.randn(N, J * C, generator=g) / N ** 0.5
    h = torch.tanh(torch.randn(ndst, C, generator=g))
    dx = torch.full((ndst, C), 7.0, device=DEV)
    o.sparse_conv_bwd_data_scatter(dy.to(DEV), w.to(DEV), perm.to(DEV), C, h.to(DEV), dx, o.ACT_TANH)
    full = (dy.double() @ w.double()).reshape(rows, J, C)
    ref = torch.full((ndst, C), 7.0, dtype=torch.float64)
    m = perm >= 0
    ref[perm[m].long()] = full[m] * (1 - h.double()[perm[m].long()] ** 2)
    np.testing.assert_allclose(dx.cpu().numpy(), ref.numpy(), rtol=5e-6, atol=2e-6)


# ------------------------------------------------------------------------------- PointNet++ group-all level, fused
@pytest.mark.parametrize("B,proprio", [(1, 0), (37, 5), (300, 0)])
def test_sa_groupall_fused_matches_fp64(B, proprio):
    """csrc/sa_groupall.hip: last layer + max over the cloud in one kernel, and the structured backward (no dense GEMM on the
    one-non-zero-per-(cloud, channel) gradient), against the same level written in fp64 torch.  Tolerances as the Linear kernels."""
    o = ops()
    R_, CK, CO = 64, 256, 512
    assert o.sa_groupall_supported(CK, CO, R_) and not o.sa_groupall_supported(128, CO, R_) and not o.sa_groupall_supported(CK, CO, 48)
    g = torch.Generator().manual_seed(100 + B)
    h = torch.tanh(torch.randn(B * R_, CK, generator=g))
    W = torch.randn(CO, CK, generator=g) / 16.0
    bias = torch.randn(CO, generator=g) * 0.1
    dfeat_full = torch.randn(B, CO + proprio, generator=g)
    hd, Wd, bd = h.to(DEV), W.to(DEV), bias.to(DEV)
    packed = torch.empty(int(o.lib.pm_sa_groupall_packed_elems(CK, CO)), device=DEV)
    o.sa_groupall_pack(Wd, packed)
    fbuf = torch.full((B, CO + proprio), 7.0, device=DEV)
    arg = o.sa_groupall_fwd(hd, B, R_, bd, packed, fbuf[:, :CO])
    z = torch.tanh(h.double() @ W.double().t() + bias.double()).view(B, R_, CO)
    want, want_arg = z.max(dim=1)
    assert float((fbuf[:, :CO].cpu().double() - want).abs().max()) < 2e-5
    if proprio:
        assert bool((fbuf[:, CO:] == 7.0).all())                       # the columns beside the feature block are the caller's
    a = arg.cpu().long()
    assert int(a.min()) >= 0 and int(a.max()) < R_
    picked = torch.gather(z, 1, a.view(B, 1, CO)).view(B, CO)
    assert float((picked - want).abs().max()) < 2e-6                     # the chosen row attains the maximum (fp32 ties may differ)
    assert float((a == want_arg).float().mean()) > 0.999
    if B > 1:                                                            # a NaN row: torch.max semantics (NaN, the first NaN's row)
        hn = hd.clone()
        hn[R_ * 1 + 5] = float("nan")
        hn[R_ * 1 + 9, 3] = float("nan")
        fb2 = torch.empty(B, CO, device=DEV)
        arg2 = o.sa_groupall_fwd(hn, B, R_, bd, packed, fb2)
        assert bool(torch.isnan(fb2[1]).all()) and bool((arg2[1] == 5).all())
        assert torch.equal(fb2[0], fbuf[0, :CO]) and torch.equal(fb2[2:], fbuf[2:, :CO]) and torch.equal(arg2[0], arg[0])
    # backward, routed through the rows the kernel chose
    dfd = dfeat_full.to(DEV)
    dh = torch.full((B * R_, CK), float("nan"), device=DEV)
    dW = torch.full((CO, CK), float("nan"), device=DEV)
    db = torch.full((CO,), float("nan"), device=DEV)
    ws = o.Workspace(torch.device(DEV))
    o.sa_groupall_bwd(dfd[:, :CO], fbuf[:, :CO], arg, Wd, hd, B, R_, dh, dW, db, ws)
    y = fbuf[:, :CO].cpu().double()
    dz = dfeat_full[:, :CO].double() * (1.0 - y * y)
    D = torch.zeros(B, R_, CO, dtype=torch.float64)
    D.scatter_(1, a.view(B, 1, CO), dz.view(B, 1, CO))
    h3 = h.double().view(B, R_, CK)
    dh_ref = (D @ W.double()) * (1.0 - h3 * h3)
    dW_ref = torch.einsum("brc,brk->ck", D, h3)
    assert rel_err(dh.view(B, R_, CK), dh_ref) < 1e-5
    assert rel_err(dW, dW_ref) < 1e-5
    assert rel_err(db, dz.sum(0)) < 1e-5
    # run-to-run identical (fixed summation orders)
    dh2, dW2, db2 = torch.empty_like(dh), torch.empty_like(dW), torch.empty_like(db)
    o.sa_groupall_bwd(dfd[:, :CO], fbuf[:, :CO], arg, Wd, hd, B, R_, dh2, dW2, db2, ws)
    assert torch.equal(dh, dh2) and torch.equal(dW, dW2) and torch.equal(db, db2)


def test_pointnet2_fused_groupall_equals_the_unfused_level():
    """The whole PointNet2 plug-in with and without the fused group-all level: same outputs and parameter gradients (the
    two differ in summation order only)."""
    from partmanip_amd.algo_utils import ActorCritic
    B, P, C, A = 6, 1024, 3, 5
    shape = dict(npoints=[128, 64], radii=[0.3, 0.6], nsamples=[32, 32], mlps=[[64, 64, 128], [128, 128, 256], [256, 512]])
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(B, P * C, generator=g) * 2 - 1).to(DEV)
    dy = torch.randn(B, A, generator=g).to(DEV)
    outs, grads = [], []
    for fused, direct in ((True, True), (False, False), (True, False)):     # (direct: the last SA level writes into the group-all rows)
        torch.manual_seed(11)
        net = dict(name="PointNet2", activation="tanh", fused_groupall=fused, groupall_direct_rows=direct, **shape)
        ac = ActorCritic(P * C, A, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net), 0).to(DEV)
        assert ac.actor._ga_fused == fused and ac.actor._ga_direct == direct
        f = ac.flat()
        out = ac.actor.hip_forward(x)
        ac.actor.hip_backward(dy)
        outs.append(out.clone())
        grads.append(f["grad_actor"].clone())
    for i in (0, 2):
        assert rel_err(outs[i], outs[1].cpu()) < 2e-5
        off = 0
        for k, v in ac.actor.named_parameters():
            a_, b_ = grads[i][off:off + v.numel()], grads[1][off:off + v.numel()]
            off += v.numel()
            assert rel_err(a_, b_.cpu()) < 2e-4, (i, k)


def test_sa_groupall_at_the_bench_size_equals_linear_plus_max_pool():
    """2048 clouds x 64 rows, 256 -> 512 (the shape `bench.py --workload vision_pn2` runs): the fused level against the launches it
    replaces -- Linear + max-pool forward; pooled-gradient scatter + the two dense GEMMs backward -- on the same device tensors."""
    o = ops()
    B, R_, CK, CO = 2048, 64, 256, 512
    g = torch.Generator(device=DEV).manual_seed(9)
    h = torch.tanh(torch.randn(B * R_, CK, device=DEV, generator=g))
    W = torch.randn(CO, CK, device=DEV, generator=g) / 16.0
    bias = torch.randn(CO, device=DEV, generator=g) * 0.1
    dfeat = torch.randn(B, CO, device=DEV, generator=g)
    ws = o.Workspace(torch.device(DEV))
    packed = torch.empty(int(o.lib.pm_sa_groupall_packed_elems(CK, CO)), device=DEV)
    o.sa_groupall_pack(W, packed)
    feat = torch.empty(B, CO, device=DEV)
    arg = o.sa_groupall_fwd(h, B, R_, bias, packed, feat)
    y = torch.empty(B * R_, CO, device=DEV)
    o.linear_fwd(h, W, bias, y, o.ACT_TANH)
    feat2 = torch.empty(B, CO, device=DEV)
    arg2 = o.maxpool_rows(y, B, R_, feat2)
    assert float((feat - feat2).abs().max()) < 2e-5
    assert float((arg != arg2).float().mean()) < 2e-3                    # near-ties only (two summation orders)
    dh, dW, db = torch.empty_like(h), torch.empty_like(W), torch.empty_like(bias)
    o.sa_groupall_bwd(dfeat, feat, arg, W, h, B, R_, dh, dW, db, ws)
    dy = o.maxpool_rows_bwd(dfeat, arg, R_, y_tanh=y)                   # routed through the fused kernel's rows
    dW2, db2, dh2 = torch.empty_like(W), torch.empty_like(bias), torch.empty_like(h)
    o.linear_bwd_weight(dy, h, dW2, db2, ws)
    o.linear_bwd_data(dy, W, h, dh2, o.ACT_TANH)
    assert rel_err(dh, dh2.cpu()) < 2e-5 and rel_err(dW, dW2.cpu()) < 2e-5 and rel_err(db, db2.cpu()) < 2e-5
    assert bool(torch.isfinite(dh).all())
