"""The integer / copy entry points added in round 5 (the compact SparseUNet decoder backward's row bookkeeping and the PointNet++
plug-in's column-block glue), each against a plain numpy statement of its contract in include/partmanip_hip.h.  Integer outputs
bit-exact; `pm_child_sum_f32` / `pm_rows_gather_bwd_skip_f32` sum in a fixed order and are compared with the same order in numpy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ops():
    from partmanip_amd import ops as o
    return o


@pytest.mark.parametrize("B,S", [(1, 1), (5, 32), (130, 64), (7, 20)])
def test_rows_uniq_names_the_distinct_rows_of_every_cloud(B, S):
    o = ops()
    g = np.random.default_rng(B * 100 + S)
    P, R0 = 50, B * 50
    arg = g.integers(0, min(P, max(2, S // 2)), size=(B, S)).astype(np.int32)          # many duplicates
    u, um, rank = (t.cpu().numpy() for t in o.rows_uniq(torch.from_numpy(arg).to(DEV), S, R0, row_base=P))
    for b in range(B):
        v = arg[b] + b * P
        d = np.unique(v)
        assert np.array_equal(u[b, :len(d)], d) and np.all(u[b, len(d):] == R0)
        assert np.array_equal(um[b, :len(d)], d) and np.all(um[b, len(d):] == -1)
        assert np.array_equal(u[b][rank[b]], v)
    if S <= P:                                               # all ids distinct: no padding slot at all, rank = the sorted position
        perm = np.stack([g.permutation(P)[:S] for _ in range(B)]).astype(np.int32)
        ud, umd, rd = (t.cpu().numpy() for t in o.rows_uniq(torch.from_numpy(perm).to(DEV), S, R0, row_base=P))
        for b in range(B):
            assert np.array_equal(ud[b], np.sort(perm[b]) + b * P) and np.array_equal(umd[b], ud[b])
            assert np.array_equal(ud[b][rd[b]], perm[b] + b * P)
    # second level: through a parent table, the padding id maps to the next level's padding id
    R1 = 37
    parent = g.integers(0, R1, size=R0).astype(np.int32)
    u1, um1, rank1 = (t.cpu().numpy() for t in o.rows_uniq(torch.from_numpy(u).to(DEV), S, R1, table=torch.from_numpy(parent).to(DEV), pad_in=R0))
    for b in range(B):
        v = np.where(u[b] == R0, R1, parent[np.minimum(u[b], R0 - 1)])
        d = np.unique(v[v != R1])
        assert np.array_equal(u1[b, :len(d)], d) and np.all(u1[b, len(d):] == R1) and np.all(um1[b, len(d):] == -1)
        assert np.array_equal(u1[b][rank1[b]], v)


@pytest.mark.parametrize("B,S,C", [(3, 32, 32), (9, 64, 64), (2, 5, 8)])
def test_child_sum_adds_children_in_slot_order(B, S, C):
    o = ops()
    g = np.random.default_rng(S + C)
    x = g.standard_normal((B * S, C)).astype(np.float32)
    rank = g.integers(0, max(1, S // 3), size=(B, S)).astype(np.int32)
    y = o.child_sum(torch.from_numpy(x).to(DEV), torch.from_numpy(rank).to(DEV), torch.empty(B * S, C, device=DEV)).cpu().numpy()
    want = np.zeros_like(x)
    for b in range(B):
        for i in range(S):                                   # ascending i: the kernel's order
            want[b * S + rank[b, i]] = want[b * S + rank[b, i]] + x[b * S + i]
    assert np.array_equal(y, want)


def test_rowmap_scatter_table_rows_vcat_table_and_exclusive_scan():
    o = ops()
    g = np.random.default_rng(3)
    n, pad = 1000, 1000
    ids = g.permutation(n)[:300].astype(np.int32)
    ids[::7] = pad
    m = o.rowmap_scatter(n, torch.from_numpy(ids).to(DEV), pad).cpu().numpy()
    want = np.full(n, -1, np.int32)
    for k, r in enumerate(ids):
        if r != pad:
            want[r] = k
    assert np.array_equal(m, want)
    table = g.integers(-1, 500, size=(200, 27)).astype(np.int32)
    sel = g.integers(-1, 200, size=77).astype(np.int32)
    got = o.table_rows(torch.from_numpy(table).to(DEV), torch.from_numpy(sel).to(DEV)).cpu().numpy()
    assert np.array_equal(got, np.where(sel[:, None] >= 0, table[np.maximum(sel, 0)], -1))
    parent = g.integers(0, 40, size=123).astype(np.int32)
    vt = o.voxel_vcat_table(torch.from_numpy(parent).to(DEV), 2, 40).cpu().numpy()
    assert np.array_equal(vt, np.stack([parent * 2, parent * 2 + 1, np.arange(123) + 80], 1))
    for nn in (1, 5, 1024, 3000):
        c = g.integers(0, 900, size=nn).astype(np.int32)
        base = torch.empty(nn + 1, dtype=torch.int32, device=DEV)
        ct = torch.from_numpy(c).to(DEV)
        o.check(o.lib.pm_exclusive_scan_i32(ct.data_ptr(), nn, base.data_ptr(), base.data_ptr() + 4 * nn, torch.cuda.current_stream().cuda_stream),
                "pm_exclusive_scan_i32")
        assert np.array_equal(base.cpu().numpy(), np.concatenate([[0], np.cumsum(c)]).astype(np.int32))


def test_rows_gather_bwd_with_a_sparse_skip_equals_the_dense_accumulate_form():
    """pm_rows_gather_bwd_skip_f32 == the accumulate = 2 form on a zero-filled tensor that holds the skip rows (bit for bit)."""
    o = ops()
    g = np.random.default_rng(4)
    rows, rows_c, C = 500, 90, 16
    dcols = torch.from_numpy(g.standard_normal((rows_c, 8 * C)).astype(np.float32)).to(DEV)
    parent = torch.from_numpy(g.integers(-1, rows_c, size=(rows, 1)).astype(np.int32)).to(DEV)
    slot = torch.from_numpy(g.integers(0, 8, size=(rows, 1)).astype(np.int32)).to(DEV)
    y = torch.from_numpy(np.tanh(g.standard_normal((rows, C))).astype(np.float32)).to(DEV)
    ids = torch.from_numpy(g.permutation(rows)[:60].astype(np.int32)).to(DEV)
    vals = torch.from_numpy(g.standard_normal((60, C)).astype(np.float32)).to(DEV)
    dense = torch.zeros(rows, C, device=DEV)
    dense[ids.long()] = vals
    a = o.rows_gather_bwd(dcols, parent, C, dense.clone(), tslot=slot, mode=1, y_tanh=y, accumulate=2)
    b = o.rows_gather_bwd(dcols, parent, C, torch.empty(rows, C, device=DEV), tslot=slot, mode=1, y_tanh=y,
                          skip=(vals, o.rowmap_scatter(rows, ids, -7)))
    assert torch.equal(a, b)


def test_col_blocks_copies_permuted_blocks_and_zero_fills():
    o = ops()
    g = np.random.default_rng(5)
    src = g.standard_normal((64, 260)).astype(np.float32)
    dst = torch.full((64, 288), 7.0, device=DEV)
    o.col_blocks(dst, torch.from_numpy(src).to(DEV), [(3, 259, 0), (0, 3, 256)])
    want = np.zeros((64, 288), np.float32)
    want[:, :256], want[:, 256:259] = src[:, 3:259], src[:, :3]
    assert np.array_equal(dst.cpu().numpy(), want)
    # a strided destination view, columns below col0 untouched, no zero fill
    big = torch.full((10, 40), 3.0, device=DEV)
    view = big[:, 4:36]
    o.col_blocks(view, torch.from_numpy(src[:10, :8].copy()).to(DEV), [(0, 8, 20)], zero_other=False, col0=16)
    w = np.full((10, 40), 3.0, np.float32)
    w[:, 4 + 20:4 + 28] = src[:10, :8]
    assert np.array_equal(big.cpu().numpy(), w)
    o.col_blocks(view, torch.from_numpy(src[:10, :8].copy()).to(DEV), [], col0=30)
    w[:, 4 + 30:36] = 0.0
    assert np.array_equal(big.cpu().numpy(), w)
    # the zero-only form needs no source; a block that would land below col0 (silently not copied before), overlapping
    # destination blocks, a block past the destination's columns and an in-place call touching what it writes are REFUSED
    o.col_blocks(view, None, [], col0=28)
    w[:, 4 + 28:36] = 0.0
    assert np.array_equal(big.cpu().numpy(), w)
    s8 = torch.from_numpy(src[:10, :8].copy()).to(DEV)
    for bad in (dict(blocks=[(0, 8, 10)], col0=16),                    # destination starts below col0
                dict(blocks=[(0, 8, 20), (0, 4, 24)], col0=16),        # the two blocks overlap in the destination
                dict(blocks=[(0, 8, 28)], col0=16)):                   # runs past the last column
        with pytest.raises(RuntimeError):
            o.col_blocks(view, s8, **bad)
    with pytest.raises(RuntimeError):
        o.col_blocks(view, view, [(16, 24, 20)], col0=16)              # in place, reading what it writes
    before = big.clone()
    o.col_blocks(view, view, [(0, 8, 20)], zero_other=False, col0=16)  # in place, reading below col0: legal
    before[:, 4 + 20:4 + 28] = before[:, 4:12]
    assert torch.equal(big, before)


def test_gather_copy_and_the_pointnet2_weight_arena_equal_the_per_copy_entry_points():
    """pm_gather_copy_f32 (dst[q] = src[table[q]], 0 where table[q] < 0) and its use: every weight-derived operand copy of a
    PointNet2 network (two pm_sa_pack_weights_f32 outputs, the aligned W1 feature block, the consumer's operand copy, the group-all
    pack, the K-step-padded first group-all layer) refreshed by ONE gather whose table was recorded by running those entry points on
    index-valued weights.  Outputs, arg-max tables and every parameter gradient of a forward + backward must be BIT-identical to the
    per-copy path (`weight_arena: False`), before and after the parameters change in place (an optimiser step), and the arena must
    hold exactly what the entry points write."""
    o = ops()
    g = torch.Generator(device=DEV).manual_seed(3)
    src = torch.randn(1000, device=DEV, generator=g)
    table = torch.randint(-1, 1000, (5000,), device=DEV, generator=g, dtype=torch.int32)
    dst = torch.full((5000,), 9.0, device=DEV)
    o.gather_copy(dst, src, table)
    want = torch.where(table >= 0, src[table.clamp(min=0).long()], torch.zeros(()).to(DEV))
    assert torch.equal(dst, want)

    from partmanip_amd.algo_utils import ActorCritic
    from tests.golden import cases
    from tests.helpers import t
    B, P, A = 24, 1024, 6
    x = (torch.rand(B, P, 3, device=DEV, generator=g) * 2 - 1).reshape(B, 3 * P)
    dy = torch.randn(B, A, device=DEV, generator=g)
    res = {}
    for arena in (True, False):
        net = dict(name="PointNet2", activation="tanh", weight_arena=arena)
        sd = cases.actor_critic_state(net, 3 * P, A, 0.5, 21)
        ac = ActorCritic(3 * P, A, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
        ac.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
        f = ac.flat()
        outs = []
        for step in range(2):
            out = ac.actor.hip_forward(x)
            assert (ac.actor._arena_views is not None) == arena
            ac.actor.hip_backward(dy)
            outs.append((out.clone(), [s_[1].clone() for s_ in ac.actor._saved], f["grad_actor"][:f["n_actor"]].clone()))
            if arena:                                          # the arena holds exactly what the entry points write
                views = ac.actor._arena_views
                assert {k[0] for k in views} == {"sa_packed", "w1f", "dyc", "ga_packed", "ga_w0p"}
                ref = {k: torch.full_like(v, float("nan")) for k, v in views.items()}
                ac.actor._arena_fill_by_entry_points(ref)
                for k in views:
                    assert torch.equal(views[k], ref[k]), k
            f["actor"][:f["n_actor"]].add_(0.01 * torch.sin(torch.arange(f["n_actor"], device=DEV, dtype=torch.float32)))   # "an optimiser step"
        res[arena] = outs
    for (o1, a1, g1), (o2, a2, g2) in zip(res[True], res[False]):
        assert torch.equal(o1, o2) and all(torch.equal(p_, q_) for p_, q_ in zip(a1, a2)) and torch.equal(g1, g2)
    assert not torch.equal(res[True][0][0], res[True][1][0])               # the second forward did see the changed parameters
