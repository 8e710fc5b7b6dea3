"""The sparse-voxel U-Net restatement (oracle/ref_cpu.py::sparse_unet_forward) is parity-UNPINNED: the reference names the
backbone (README.md:30) but does not contain it (README.md:23).  What can be pinned is its meaning: on a FULLY OCCUPIED grid
a submanifold 3^3 convolution is torch's conv3d with padding 1, the strided level is conv3d(kernel 2, stride 2), unpooling is
nearest-neighbour upsampling -- so the whole network must equal the dense U-Net written with torch.nn.functional.  Plus the
set properties a voxel network has to have: invariance to the order of the rows and to duplicated / padding rows."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as R
from tests.golden import cases

NET = dict(name="SparseUNet", activation="tanh", point_num=64, grid=4, channels=[8, 12, 16])


def _params(net, O, A, seed, proprio=0):
    sd = cases.actor_critic_state(net, O, A, 0.5, seed, proprio)
    return {k: torch.from_numpy(v.copy()).double() for k, v in sd.items()}


def _dense_cloud(Rg, seed, shuffle=True):
    g = np.random.default_rng(seed)
    cells = np.stack(np.meshgrid(np.arange(Rg), np.arange(Rg), np.arange(Rg), indexing="ij"), -1).reshape(-1, 3)
    f = g.uniform(-0.2, 0.2, size=(cells.shape[0], 1))
    rows = np.concatenate([cells.astype(np.float64), f], 1)
    if shuffle:
        rows = rows[g.permutation(rows.shape[0])]
    return rows


def test_fully_occupied_grid_equals_the_dense_unet_in_torch():
    Rg, A = 4, 5
    net = dict(NET, point_num=Rg ** 3, grid=Rg)
    p = _params(net, 4 * Rg ** 3, A, 31)
    clouds = np.stack([_dense_cloud(Rg, 1), _dense_cloud(Rg, 2)])
    x = torch.from_numpy(clouds.reshape(2, -1))
    out = R.sparse_unet_forward(p, "actor", net, x)

    c0, c1, c2 = net["channels"]
    vol = torch.zeros(2, 4, Rg, Rg, Rg, dtype=torch.float64)           # (f, x/R, y/R, z/R) volumes
    for b in range(2):
        for row in clouds[b]:
            i, j, k = (int(v) for v in row[:3])
            vol[b, :, i, j, k] = torch.tensor([row[3], i / Rg, j / Rg, k / Rg])
    w3 = lambda n, co, ci: p[f"actor.{n}.weight"].view(co, 27, ci).permute(0, 2, 1).reshape(co, ci, 3, 3, 3)
    w2 = lambda n, co, ci: p[f"actor.{n}.weight"].view(co, 8, ci).permute(0, 2, 1).reshape(co, ci, 2, 2, 2)
    bb = lambda n: p[f"actor.{n}.bias"]
    H0 = torch.tanh(F.conv3d(vol, w3("conv0", c0, 4), bb("conv0"), padding=1))
    D1 = torch.tanh(F.conv3d(H0, w2("down0", c1, c0), bb("down0"), stride=2))
    H1 = torch.tanh(F.conv3d(D1, w3("conv1", c1, c1), bb("conv1"), padding=1))
    D2 = torch.tanh(F.conv3d(H1, w2("down1", c2, c1), bb("down1"), stride=2))
    H2 = torch.tanh(F.conv3d(D2, w3("conv2", c2, c2), bb("conv2"), padding=1))
    up = lambda v: F.interpolate(v, scale_factor=2, mode="nearest")
    lin1 = lambda n, v: torch.tanh(torch.einsum("oc,bcxyz->boxyz", p[f"actor.{n}.weight"], v) + bb(n).view(1, -1, 1, 1, 1))
    E1 = lin1("up1", torch.cat([up(H2), H1], 1))
    E0 = lin1("up0", torch.cat([up(E1), H0], 1))
    feat = E0.flatten(2).max(-1)[0]
    h = torch.tanh(F.linear(feat, p["actor.final_mlp.0.weight"], p["actor.final_mlp.0.bias"]))
    h = torch.tanh(F.linear(h, p["actor.final_mlp.2.weight"], p["actor.final_mlp.2.bias"]))
    ref = F.linear(h, p["actor.final_mlp.4.weight"], p["actor.final_mlp.4.bias"])
    assert float((out - ref).abs().max()) < 1e-9          # the restatement rounds the input features to fp32


def test_row_order_duplicates_and_padding_do_not_change_the_output():
    P, Rg, A = 64, 12, 4
    net = dict(NET, point_num=P, grid=Rg)
    p = _params(net, 4 * P, A, 32)
    x = torch.from_numpy(cases.sparse_clouds(3, P, Rg, 5, n_distinct=40, pad_tail=6)).double()
    out = R.sparse_unet_forward(p, "actor", net, x)
    perm = torch.from_numpy(np.random.default_rng(0).permutation(P))
    xp = x.view(3, P, 4)[:, perm].reshape(3, -1)
    assert float((R.sparse_unet_forward(p, "actor", net, xp) - out).abs().max()) < 1e-12
    # drop every duplicated / padding row by overwriting it with a copy of row 0 (still a duplicate): same set of voxels
    xs = x.view(3, P, 4).clone()
    xs[:, 40:] = xs[:, :1]
    xs[:, 40] = torch.tensor([0.0, 0.0, 0.0, 0.125], dtype=torch.float64)     # keep the padding voxel (0,0,0) itself
    assert float((R.sparse_unet_forward(p, "actor", net, xs.reshape(3, -1)) - out).abs().max()) < 1e-12


def test_geometry_tables_are_consistent():
    P, Rg = 48, 10
    x = cases.sparse_clouds(2, P, Rg, 6, n_distinct=30, pad_tail=4)
    g = R.sparse_unet_geometry(x, P, 4, Rg)
    n0, n1, n2 = g["rows"]
    assert n0 == 2 * P and 0 < n2 <= n1 <= n0
    # the centre offset of a canonical row is the row itself; a duplicate's centre is its canonical twin (a lower row)
    centre = g["nbr0"][:, 13]
    assert (centre <= np.arange(n0)).all() and (centre >= 0).all()
    canon = centre == np.arange(n0)
    assert ((g["l1"]["parent_canon"] >= 0) == canon).all()
    # every canonical fine row is the child of its parent in its slot, and nothing else is
    ch = g["l1"]["child"]
    for r in np.nonzero(canon)[0]:
        assert ch[g["l1"]["parent"][r], g["l1"]["slot"][r]] == r
    assert (ch >= 0).sum() == canon.sum()
    # mirrored neighbour tables: o is the neighbour of r  <=>  r is the (26 - o) neighbour of that row
    nb = g["nbr1"]
    for r in range(n1):
        for o in range(27):
            if nb[r, o] >= 0:
                assert nb[nb[r, o], 26 - o] == r


@pytest.mark.parametrize("dups", [False, True])
def test_batched_torch_geometry_equals_the_dict_form(dups):
    """`sparse_unet_geometry_torch` (dense index grids, batched: what the whole-update parity tests run on the GPU through ATen) builds
    the tables of the per-cloud dict form entry for entry -- canonical rows of duplicated coordinates, out-of-grid taps, odd grid sizes."""
    P, Rg = 96, 13
    x = cases.sparse_clouds(5, P, Rg, 11, n_distinct=60 if dups else None, pad_tail=7 if dups else 0)
    a, b = R.sparse_unet_geometry(x, P, 4, Rg), R.sparse_unet_geometry_torch(torch.from_numpy(x), P, 4, Rg)
    assert a["rows"] == b["rows"]
    assert np.array_equal(a["feat0"], b["feat0"].numpy())
    for k in ("nbr0", "nbr1", "nbr2"):
        assert np.array_equal(a[k], b[k].numpy()), k
    for lv in ("l1", "l2"):
        for k in ("child", "parent", "parent_canon", "slot"):
            assert np.array_equal(a[lv][k], b[lv][k].numpy().reshape(a[lv][k].shape)), (lv, k)
        assert np.array_equal(np.concatenate(a[lv]["coords"]), b[lv]["coords"].numpy())
