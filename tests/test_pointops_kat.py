"""Known-answer vectors for FPS / ball query / grouping (tests/golden/pointops_kat.npz).

The expected indices come from tests/golden/make_pointops_kat.py: brute force over Python `int`s on integer-lattice
clouds (every squared distance exact in fp32), written from the operators' published definitions and sharing NO code
with oracle/ref_cpu.py.  The reference's only FPS is the un-vendored pytorch3d call (utils/depth2tsdf.py:113,160) and
it has no ball query at all, so these operators stay "parity unpinned" -- but with these vectors neither the
restatement (CPU test) nor the HIP kernels (GPU test) are checked against themselves only: a shared misreading of the
tie-break (lowest index), of `<` against `<=` r^2, of the padding rules or of K > P would fail here.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

KAT = np.load(os.path.join(os.path.dirname(__file__), "golden", "pointops_kat.npz"))
FPS_CASES = ["fps_small", "fps_dups", "fps_k_gt_p", "fps_d4", "fps_wave_max", "fps_workgroup", "fps_streaming", "fps_voxel_grid"]
BQ_CASES = ["bq_r5", "bq_r3", "bq_dense", "bq_wide"]
DEV = "cuda:0"


def _f32(a):
    return np.ascontiguousarray(a.astype(np.float32))


# --------------------------------------------------------------------------------- the CPU restatement
@pytest.mark.parametrize("name", FPS_CASES)
def test_oracle_fps_matches_the_known_answers(name):
    idx = R.fps(_f32(KAT[name + "_xyz"]), int(KAT[name + "_K"]))
    assert np.array_equal(idx, KAT[name + "_idx"])


def test_oracle_varlen_fps_matches_the_known_answers():
    idx = R.fps(_f32(KAT["varlen_xyz"]), int(KAT["varlen_K"]), KAT["varlen_lengths"])
    assert np.array_equal(idx, KAT["varlen_idx_pad"])


@pytest.mark.parametrize("name", BQ_CASES)
def test_oracle_ball_query_and_grouping_match_the_known_answers(name):
    xyz, ctr = _f32(KAT[name + "_xyz"]), _f32(KAT[name + "_centers"])
    idx = R.ball_query(xyz, ctr, float(KAT[name + "_radius"]), int(KAT[name + "_nsample"]))
    assert np.array_equal(idx, KAT[name + "_idx"])
    assert np.array_equal(R.group_points(_f32(KAT[name + "_feat"]), idx), _f32(KAT[name + "_grouped"]))


def test_known_answers_contain_the_cases_they_are_for():
    """The fixture itself: exact ties, points exactly ON the sphere, empty balls, overfull balls, K > P, exhausted clouds."""
    xyz, idx = KAT["bq_r5_xyz"].astype(np.int64), KAT["bq_r5_idx"]
    ctr = KAT["bq_r5_centers"].astype(np.int64)
    d2 = ((xyz[:, None, :, :] - ctr[:, :, None, :]) ** 2).sum(-1)
    assert (d2 == 25).sum() > 50                                   # lattice points exactly at the radius: excluded by `<`
    assert not np.any(np.take_along_axis(d2, idx.astype(np.int64), axis=2)[:, :-3] >= 25)
    assert np.all(idx[:, -3] == 0)                                 # the far-away centre: an empty ball is all zeros
    assert (KAT["bq_dense_idx"][0, :, -1] != KAT["bq_dense_idx"][0, :, 0]).any()     # more hits than nsample somewhere
    assert (KAT["fps_k_gt_p_idx"] < 0).sum() == 2 * 10 and (KAT["fps_dups_idx"][:, 12:] == 0).all()
    a = KAT["fps_small_xyz"].astype(np.int64)[0]
    d0 = ((a - a[0]) ** 2).sum(-1)
    assert len(np.unique(d0)) < 0.75 * len(d0)                       # exact ties are the rule on a lattice (131 distinct of 200)
    assert np.array_equal(KAT["varlen_idx_pad"][0], -np.ones(48, dtype=np.int64))     # an empty cloud


# --------------------------------------------------------------------------------- the HIP kernels
def _ops():
    from partmanip_amd import ops
    return ops


@pytest.mark.gpu
@pytest.mark.parametrize("name", FPS_CASES)
def test_hip_fps_matches_the_known_answers(name):
    o = _ops()
    xyz = torch.from_numpy(_f32(KAT[name + "_xyz"])).to(DEV)
    idx = o.fps(xyz, int(KAT[name + "_K"]), o.Workspace(torch.device(DEV)))
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), KAT[name + "_idx"])


@pytest.mark.gpu
@pytest.mark.parametrize("pad", [True, False])
def test_hip_varlen_fps_matches_the_known_answers(pad):
    o = _ops()
    xyz = torch.from_numpy(_f32(KAT["varlen_xyz"])).to(DEV)
    n = torch.from_numpy(KAT["varlen_lengths"]).to(DEV)
    idx = o.fps_varlen(xyz, n, int(KAT["varlen_K"]), o.Workspace(torch.device(DEV)), pad=pad)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), KAT["varlen_idx_pad" if pad else "varlen_idx_nopad"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", BQ_CASES)
def test_hip_ball_query_and_grouping_match_the_known_answers(name):
    o = _ops()
    xyz = torch.from_numpy(_f32(KAT[name + "_xyz"])).to(DEV)
    ctr = torch.from_numpy(_f32(KAT[name + "_centers"])).to(DEV)
    idx = o.ball_query(xyz, ctr, float(KAT[name + "_radius"]), int(KAT[name + "_nsample"]))
    assert np.array_equal(idx.cpu().numpy(), KAT[name + "_idx"])
    feat = torch.from_numpy(_f32(KAT[name + "_feat"])).to(DEV)
    assert np.array_equal(o.group_points(feat, idx).cpu().numpy(), _f32(KAT[name + "_grouped"]))
    dout = torch.from_numpy(_f32(KAT[name + "_dout"])).to(DEV)
    dfeat = o.group_points_bwd(dout, idx, xyz.shape[1])            # small-integer sums: exact in fp32 in any order
    assert np.array_equal(dfeat.cpu().numpy(), _f32(KAT[name + "_dfeat"]))
