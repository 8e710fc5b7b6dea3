"""The algorithmic claim behind the round-4 set-abstraction kernels, checked on the CPU restatement alone (no HIP):

    a PointNet++ level evaluated over each ball-query group's DISTINCT rows equals the level over the padded 32-row groups

-- ball query pads a short group with copies of its first hit (oracle/ref_cpu.py::ball_query, pinned by the known-answer
vectors of tests/test_pointops_kat.py); a copy is the same source point against the same centre as row 0, so it computes the
same activations, and `max` over a group returns the LOWEST row attaining the maximum: a copy never wins and never receives a
gradient.  The HIP kernels are compared with their padded twins bit for bit on the GPU
(tests/test_gpu_kernels.py::test_sa_packed_rows_equal_the_dense_level); this file states the same equivalence for the
restatement in fp64: outputs, arg-max rows and every parameter gradient.
"""
import numpy as np
import torch

from oracle import ref_cpu as R

NET = dict(name="PointNet2", activation="tanh", point_num=256, npoints=[48, 12], radii=[0.35, 0.7], nsamples=[32, 32],
           mlps=[[16, 16, 24], [24, 24, 32], [32, 40]])


def _params(net, C, A, seed):
    g = torch.Generator().manual_seed(seed)
    p, cf = {}, C - 3
    for l, dims in enumerate(net["mlps"]):
        cin = R._pad4(3 + cf)
        for i, d in enumerate(dims):
            p[f"actor.sa.{l}.{2 * i}.weight"] = (torch.randn(d, cin, generator=g, dtype=torch.float64) / cin ** 0.5).requires_grad_(True)
            p[f"actor.sa.{l}.{2 * i}.bias"] = (torch.randn(d, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
            cin = d
        cf = dims[-1]
    for k, (o, i) in (("0", (128, cf)), ("2", (32, 128)), ("4", (A, 32))):
        p[f"actor.final_mlp.{k}.weight"] = (torch.randn(o, i, generator=g, dtype=torch.float64) / i ** 0.5).requires_grad_(True)
        p[f"actor.final_mlp.{k}.bias"] = (torch.randn(o, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    return p


def _forward_distinct_rows(p, net, x):
    """The structure of R.pointnet2_forward with every fused-able level evaluated on the distinct rows of each group only."""
    P = net["point_num"]
    B, C = x.shape[0], x.shape[1] // P
    pts = x.reshape(B, P, C)
    xyz, feat = pts[..., :3], (pts[..., 3:] if C > 3 else None)
    lin = lambda l, i, h: torch.tanh(h @ p[f"actor.sa.{l}.{2 * i}.weight"].t() + p[f"actor.sa.{l}.{2 * i}.bias"])
    args, rows_seen, rows_padded = [], 0, 0
    for l, (S, ns) in enumerate(zip(net["npoints"], net["nsamples"])):
        idx_c = torch.from_numpy(R.fps(xyz.detach().float().numpy(), S))
        centers = torch.gather(xyz, 1, idx_c.unsqueeze(-1).expand(B, S, 3))
        idx_g = R.ball_query(xyz.detach().float().numpy(), centers.detach().float().numpy(), net["radii"][l], ns)
        pooled, arg = [], []
        for b in range(B):
            for s in range(S):
                row = idx_g[b, s]
                keep = np.concatenate([row[:1], row[1:][row[1:] != row[0]]])          # entries equal to entry 0 are padding
                rows_seen += len(keep)
                rows_padded += ns
                k = torch.from_numpy(keep.astype(np.int64))
                cols = [xyz[b, k] - centers[b, s]]
                if feat is not None:
                    cols.append(feat[b, k])
                h = torch.cat(cols, dim=-1)
                pad = R._pad4(h.shape[1]) - h.shape[1]
                if pad:
                    h = torch.cat([h, torch.zeros(len(keep), pad, dtype=h.dtype)], dim=-1)
                for i in range(len(net["mlps"][l])):
                    h = lin(l, i, h)
                v, a = h.max(dim=0)
                pooled.append(v)
                arg.append(a)
        xyz, feat = centers, torch.stack(pooled).reshape(B, S, -1)
        args.append(torch.stack(arg))
    l = len(net["npoints"])
    h = torch.cat([xyz, feat], dim=-1)
    pad = R._pad4(h.shape[-1]) - h.shape[-1]
    if pad:
        h = torch.cat([h, torch.zeros(B, h.shape[1], pad, dtype=h.dtype)], dim=-1)
    for i in range(len(net["mlps"][l])):
        h = lin(l, i, h)
    f = h.max(dim=1).values
    for k in ("0", "2"):
        f = torch.tanh(f @ p[f"actor.final_mlp.{k}.weight"].t() + p[f"actor.final_mlp.{k}.bias"])
    return f @ p["actor.final_mlp.4.weight"].t() + p["actor.final_mlp.4.bias"], args, rows_seen / rows_padded


def test_distinct_rows_of_a_group_carry_the_whole_level():
    B, C, A = 3, 5, 6
    g = torch.Generator().manual_seed(12)
    x = (torch.rand(B, NET["point_num"], C, generator=g, dtype=torch.float64) * 2 - 1).reshape(B, -1)
    p = _params(NET, C, A, 5)
    out_pad, aux, args_pad = R.pointnet2_forward(p, "actor", NET, x, 0, return_aux=True)
    out_dis, args_dis, frac = _forward_distinct_rows(p, NET, x)
    assert 0.05 < frac < 0.6, frac                            # the case must actually contain padding (and not only padding)
    np.testing.assert_allclose(out_dis.detach().numpy(), out_pad.detach().numpy(), rtol=0, atol=1e-12)
    for a, b in zip(args_dis, args_pad[:len(args_dis)]):      # the winner's row inside its group is the same row
        assert torch.equal(a, b)
    w = torch.randn(B, A, generator=g, dtype=torch.float64)
    names = list(p)
    g_pad = torch.autograd.grad((out_pad * w).sum(), [p[k] for k in names])
    g_dis = torch.autograd.grad((out_dis * w).sum(), [p[k] for k in names])
    for k, a, b in zip(names, g_dis, g_pad):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=1e-11 * max(1.0, float(b.abs().max())), err_msg=k)


def test_batched_torch_fps_and_ball_query_equal_the_loop_forms():
    """`fps_torch` / `ball_query_torch` (what the whole-update parity tests run on the GPU through ATen, tests/test_gpu_wholeupdate.py)
    are the loop restatements element for element: float clouds, exact ties (duplicated points, integer voxel coordinates), short and
    empty groups, K == P."""
    rng = np.random.default_rng(0)
    for B, P, S in ((3, 200, 50), (2, 64, 64), (2, 1024, 256)):
        pts = rng.uniform(-1, 1, (B, P, 3)).astype(np.float32)
        pts[0, 5] = pts[0, 3]
        a = R.fps(pts, S)
        assert np.array_equal(a, R.fps_torch(torch.from_numpy(pts), S).numpy())
        ctr = np.take_along_axis(pts, a[..., None].repeat(3, -1), 1)
        for r, ns in ((0.2, 32), (0.05, 8), (2.5, 16)):
            want = R.ball_query(pts, ctr, r, ns)
            assert np.array_equal(want, R.ball_query_torch(torch.from_numpy(pts), torch.from_numpy(ctr), r, ns, chunk=2).numpy()), (r, ns)
    ip = rng.integers(0, 6, (3, 100, 3)).astype(np.float32)
    assert np.array_equal(R.fps(ip, 40), R.fps_torch(torch.from_numpy(ip), 40).numpy())
    far = np.full((1, 4, 3), 9.0, np.float32)                        # no point within the radius: all-zero rows
    assert not R.ball_query_torch(torch.from_numpy(ip[:1]), torch.from_numpy(far), 0.5, 8).any()


def test_forward_with_prebuilt_geometry_equals_the_inline_form():
    x = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, (3, NET["point_num"] * 3))).double()
    p = _params(NET, 3, 5, 1)
    geom = R.pointnet2_geometry(x, NET)
    a = R.pointnet2_forward(p, "actor", NET, x)
    b = R.pointnet2_forward(p, "actor", NET, x, geom=geom)
    assert torch.equal(a, b)
