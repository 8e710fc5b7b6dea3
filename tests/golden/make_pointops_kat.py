"""Known-answer vectors for the point-set operators (FPS, ball query, grouping): tests/golden/pointops_kat.npz.

    python tests/golden/make_pointops_kat.py

INDEPENDENT of oracle/ref_cpu.py on purpose: nothing is imported from the repo, no numpy arithmetic produces an
expected value.  Every cloud lives on an integer lattice (|coordinate| <= 64), so every squared distance is an
integer < 2^24 -- exact in fp32 whatever the summation order or FMA contraction -- and the expected indices are
computed with Python `int`s by brute force from the operators' published definitions:

* FPS (pytorch3d `sample_farthest_points`, defaults of utils/depth2tsdf.py:113,160: `random_start_point=False`):
  first pick = index 0; every point keeps the minimum squared distance to the picks so far; next pick = the point
  with the LARGEST such minimum, the LOWEST index among equals; a cloud shorter than K is padded with -1 (`pad`),
  or -- the depth2pc form, where the zeroed out-of-crop points stay candidates -- keeps picking (all minima are 0,
  so index 0 repeats).
* ball query (PointNet++ `query_ball_point`): the first `nsample` indices, ascending, with squared distance
  STRICTLY below radius^2; short rows padded with their first hit; an empty ball is all zeros.
* grouping: out[b, s, j, :] = feat[b, idx[b, s, j], :]; backward = the index-add of the upstream gradient.

The lattice makes exact ties the rule rather than the exception (hundreds per cloud), which is where a shared
misreading of the tie-break, of `<` against `<=`, or of the padding rule would show.
"""
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def lattice_cloud(rng, n, dim, lo, hi, distinct=None):
    if distinct is None:
        return [[rng.randint(lo, hi) for _ in range(dim)] for _ in range(n)]
    base = [[rng.randint(lo, hi) for _ in range(dim)] for _ in range(distinct)]
    return [list(base[rng.randrange(distinct)]) for _ in range(n)]


def sqdist(p, q):
    return sum((a - b) * (a - b) for a, b in zip(p, q))


def fps_int(cloud, n, K, pad):
    """cloud: list of integer points, only the first n are candidates."""
    out = [-1] * K
    if n == 0:
        return out
    mind = [None] * n                                   # None = +infinity
    sel = 0
    for j in range(K):
        if j >= n and pad:
            break
        out[j] = sel
        best, arg = -1, 0
        ps = cloud[sel]
        for i in range(n):
            d = sqdist(cloud[i], ps)
            if mind[i] is None or d < mind[i]:
                mind[i] = d
            if mind[i] > best:                          # strict: the lowest index wins a tie
                best, arg = mind[i], i
        sel = arg
    return out


def ball_query_int(cloud, centre, r2, nsample):
    hits = []
    for i, p in enumerate(cloud):
        if sqdist(p, centre) < r2:                      # strictly inside
            hits.append(i)
            if len(hits) == nsample:
                break
    if not hits:
        return [0] * nsample
    return hits + [hits[0]] * (nsample - len(hits))


def main():
    rng = random.Random(20260930)
    out = {}

    # ---- FPS, fixed length --------------------------------------------------------------------------------------
    # (name, B, P, D, K, lo, hi, distinct): the P ranges cover the one-wave (<= 2048), work-group (<= 8192) and
    # streaming kernels; `distinct` forces fewer distinct points than K (minima all 0 -> index 0 repeats)
    fps_cases = [
        ("fps_small", 3, 200, 3, 64, 0, 15, None),
        ("fps_dups", 2, 50, 3, 50, 0, 63, 10),
        ("fps_k_gt_p", 2, 40, 3, 50, -8, 8, None),
        ("fps_d4", 2, 300, 4, 40, 0, 31, None),
        ("fps_wave_max", 2, 2048, 3, 96, 0, 63, None),
        ("fps_workgroup", 2, 3000, 3, 64, -32, 32, None),
        ("fps_streaming", 1, 9000, 3, 48, 0, 64, None),
        ("fps_voxel_grid", 1, 1000, 3, 128, 0, 9, None),       # replaced below by the full 10^3 grid in row-major order
    ]
    for name, B, P, D, K, lo, hi, distinct in fps_cases:
        clouds = [lattice_cloud(rng, P, D, lo, hi, distinct) for _ in range(B)]
        if name == "fps_voxel_grid":                             # `sparse_voxel` feeds integer voxel coordinates
            clouds = [[[x, y, z] for x in range(10) for y in range(10) for z in range(10)]]
        out[name + "_xyz"] = np.array(clouds, dtype=np.int32)
        out[name + "_K"] = np.int64(K)
        out[name + "_idx"] = np.array([fps_int(c, P, K, pad=True) for c in clouds], dtype=np.int64)

    # ---- FPS, variable length (pm_fps_varlen_f32: depth2pc compaction / sparse_voxel) -------------------------------
    B, ld, K = 6, 260, 48
    clouds = [lattice_cloud(rng, ld, 3, 0, 49) for _ in range(B)]
    lengths = [0, 1, 7, 48, 130, 260]
    out["varlen_xyz"] = np.array(clouds, dtype=np.int32)
    out["varlen_lengths"] = np.array(lengths, dtype=np.int32)
    out["varlen_K"] = np.int64(K)
    out["varlen_idx_pad"] = np.array([fps_int(c, n, K, pad=True) for c, n in zip(clouds, lengths)], dtype=np.int64)
    out["varlen_idx_nopad"] = np.array([fps_int(c, n, K, pad=False) for c, n in zip(clouds, lengths)], dtype=np.int64)

    # ---- ball query + grouping -------------------------------------------------------------------------------------------
    # radius 5 on a lattice: the 3-4-5 / 0-0-5 / 0-3-4 neighbours sit EXACTLY on the sphere and must be excluded;
    # radius 3: d^2 = 9 likewise (1-2-2).  Centres: some cloud points, some off-lattice-free integer points far
    # away (empty balls), one in a dense corner (more than nsample hits).
    bq_cases = [("bq_r5", 2, 600, 40, 5, 16, 0, 12), ("bq_r3", 2, 333, 25, 3, 8, 0, 7), ("bq_dense", 1, 512, 12, 6, 32, 0, 5),
                ("bq_wide", 1, 100, 6, 64, 64, -20, 20)]
    for name, B, P, S, r, ns, lo, hi in bq_cases:
        clouds, centres, idx = [], [], []
        for b in range(B):
            c = lattice_cloud(rng, P, 3, lo, hi)
            ctr = [list(c[rng.randrange(P)]) for _ in range(S - 3)]
            ctr.append([hi + 40, hi + 40, hi + 40])             # empty ball
            ctr.append([lo, lo, lo])
            ctr.append([(lo + hi) // 2, (lo + hi) // 2, (lo + hi) // 2])
            clouds.append(c)
            centres.append(ctr)
            idx.append([ball_query_int(c, q, r * r, ns) for q in ctr])
        out[name + "_xyz"] = np.array(clouds, dtype=np.int32)
        out[name + "_centers"] = np.array(centres, dtype=np.int32)
        out[name + "_radius"] = np.int64(r)
        out[name + "_nsample"] = np.int64(ns)
        out[name + "_idx"] = np.array(idx, dtype=np.int32)
        # grouping of small-integer features and the index-add of small-integer upstream gradients (sums stay exact)
        C = 5
        feat = [[[rng.randint(-9, 9) for _ in range(C)] for _ in range(P)] for _ in range(B)]
        dout = [[[[rng.randint(-3, 3) for _ in range(C)] for _ in range(ns)] for _ in range(S)] for _ in range(B)]
        grouped = [[[[feat[b][i][k] for k in range(C)] for i in idx[b][s]] for s in range(S)] for b in range(B)]
        dfeat = [[[0] * C for _ in range(P)] for _ in range(B)]
        for b in range(B):
            for s in range(S):
                for j, i in enumerate(idx[b][s]):
                    for k in range(C):
                        dfeat[b][i][k] += dout[b][s][j][k]
        out[name + "_feat"] = np.array(feat, dtype=np.int32)
        out[name + "_grouped"] = np.array(grouped, dtype=np.int32)
        out[name + "_dout"] = np.array(dout, dtype=np.int32)
        out[name + "_dfeat"] = np.array(dfeat, dtype=np.int32)

    path = os.path.join(HERE, "pointops_kat.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(out), "arrays")
    for k in sorted(out):
        if k.endswith("_idx") or k.startswith("varlen_idx"):
            a = out[k]
            print(f"  {k}: shape {a.shape}, -1 pads {int((a < 0).sum())}")


if __name__ == "__main__":
    main()
