"""Randomised sweep of RANSAC scoring against the CPU restatement (reference submodules/solve/RANSAC.cpp:146-270,
eightPoint.cpp:225-372, solve.hpp:85-124): correspondence counts, inlier ratios, noise, intrinsics, error
bounds, sample tables and the library's scoring paths drawn from a seeded generator.  Bar: identical per-row inlier counts, bit-identical
per-row error sums, identical winner, iteration count and inlier list."""
import os

import numpy as np
import pytest

from sfm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(int(os.environ.get("SFM_FUZZ_CASES", "32"))))
def test_random_case(ctx, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([8, 9, 17, 64, 255, 256, 257, 600, 1500]))
    H = int(rng.choice([1, 63, 64, 65, 256, 300, 700]))
    K = synth.default_K()
    if rng.random() < 0.3:  # other intrinsics: skew, unequal focal lengths, off-centre principal point
        K = np.array([[rng.uniform(500, 4000), rng.uniform(-5, 5), rng.uniform(100, 3000)],
                      [0.0, rng.uniform(500, 4000), rng.uniform(100, 3000)], [0.0, 0.0, 1.0]])
    tv = synth.two_view(n, inlier_ratio=float(rng.choice([0.15, 0.5, 0.9, 1.0])), seed=int(rng.integers(1 << 30)),
                        noise_px=float(rng.choice([0.0, 0.3, 2.0])), K=K)
    x, y = tv["x"].copy(), tv["y"].copy()
    if rng.random() < 0.3:  # degenerate rows: repeated points, a point at the origin, a far-away pixel
        x[rng.integers(n)] = x[0]
        y[rng.integers(n)] = 0.0
        x[rng.integers(n)] *= 50.0
    S = synth.sample_table(n, H, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.2:
        S[rng.integers(H)] = S[0]  # a repeated sample row: the earlier one must win ties
    thr = float(rng.choice([1.0, 10.0, 0.05, 1e-6, 1e4]))
    adaptive = bool(rng.integers(2))
    loss = int(rng.choice([0, 0, 1]))  # the reference's reprojection loss, or the Sampson distance (extension)
    # Every path of the scoring: no pre-test, the packed fp32 test, the test on the matrix cores with prepared
    # operands or with operands made by the blocks (unsplit launches only), split over correspondences or not.
    ctx.set_option("ransac_filter", int(rng.integers(2)))
    ctx.set_option("ransac_split", int(rng.integers(2)))
    ctx.set_option("ransac_mfma", int(rng.integers(2)))
    ctx.set_option("ransac_mfma_ops_gib", int(rng.choice([0, 64])))
    try:
        got = ctx.ransac_pair(x, y, K, S, max_sq_err=thr, success_prob=0.999, adaptive=adaptive, loss=loss)
    finally:
        ctx.set_option("ransac_filter", 1)
        ctx.set_option("ransac_split", 1)
        ctx.set_option("ransac_mfma", 1)
        ctx.set_option("ransac_mfma_ops_gib", 64)
    exp = oracle.ransac(x, y, K, S, max_sq_err=thr, success_prob=0.999, adaptive=adaptive, loss_kind=loss)
    assert np.array_equal(got["counts"], exp["counts"]), "per-row inlier counts differ"
    assert np.array_equal(got["toterr"].view(np.uint64), exp["toterr"].view(np.uint64)), "per-row error sums differ"
    assert got["best_h"] == exp["best_h"] and got["iterations"] == exp["iterations"]
    assert np.array_equal(got["inliers"], exp["inliers"])


@pytest.mark.parametrize("seed", range(max(4, int(os.environ.get("SFM_FUZZ_CASES", "32")) // 8)))
def test_random_batch(ctx, oracle, seed):
    """Batches of pairs of unequal sizes through the batch entry point, where launches are large enough to run
    unsplit (the matrix-core pre-test and its prepared operands): winner, its count and error sum, iteration
    count and inlier list of every pair against the restatement."""
    rng = np.random.default_rng(9000 + seed)
    P = int(rng.integers(1, 20))
    H = int(rng.choice([64, 257, 520, 1024]))
    K = synth.default_K()
    if rng.random() < 0.3:
        K = np.array([[rng.uniform(500, 4000), 0.0, rng.uniform(100, 3000)], [0.0, rng.uniform(500, 4000), rng.uniform(100, 3000)],
                      [0.0, 0.0, 1.0]])
    thr = float(rng.choice([1.0, 10.0, 0.05, 1e-4, 1e3]))
    adaptive = bool(rng.integers(2))
    loss = int(rng.choice([0, 0, 1]))
    xs, ys, Ss, off = [], [], [], [0]
    for k in range(P):
        # (4096 correspondences and more: sample rows with many inliers may be handed over to the cell-by-cell kernels)
        n = int(rng.choice([0, 5, 8, 15, 16, 17, 63, 64, 65, 500, 1023, 2500, 4095, 4096, 4500, 6000, 9001],
                           p=[.03, .03, .05, .05, .05, .05, .05, .05, .05, .15, .1, .09, .05, .05, .05, .05, .05]))
        tv = synth.two_view(max(n, 8), inlier_ratio=float(rng.choice([0.15, 0.5, 0.9])), seed=int(rng.integers(1 << 30)),
                            noise_px=float(rng.choice([0.0, 0.5, 2.0])), K=K)
        x, y = tv["x"][:n].copy(), tv["y"][:n].copy()
        if n > 20 and rng.random() < 0.3:
            x[rng.integers(n)] *= 100.0
            y[rng.integers(n)] = 0.0
        xs.append(x); ys.append(y)
        Ss.append(synth.sample_table(max(n, 8), H, seed=int(rng.integers(1 << 30))) % max(n, 1))
        off.append(off[-1] + n)
    x, y, S = np.concatenate(xs), np.concatenate(ys), np.stack(Ss)
    ctx.set_option("ransac_split", int(rng.integers(2)))
    ctx.set_option("ransac_mfma", int(rng.integers(2)))
    ctx.set_option("ransac_mfma_ops_gib", int(rng.choice([0, 64])))
    ctx.set_option("ransac_defer_mib", int(rng.choice([0, 5, 2048, 2048])))  # (5 MiB: 64 rows of 9001 correspondences at a time)
    ctx.set_option("hyps_per_launch", int(rng.choice([0, 0, 64, 192])))      # rows of the first slice
    try:
        b = ctx.ransac_batch(np.array(off), x, y, K, S, H, max_sq_err=thr, success_prob=0.999, adaptive=adaptive, loss=loss)
        best = b.best()
        ioff, inl = b.inliers()
        b.close()
    finally:
        ctx.set_option("ransac_split", 1)
        ctx.set_option("ransac_mfma", 1)
        ctx.set_option("ransac_mfma_ops_gib", 64)
        ctx.set_option("ransac_defer_mib", 2048)
        ctx.set_option("hyps_per_launch", 0)
    for k in range(P):
        n = off[k + 1] - off[k]
        if n < 8:
            continue
        e = oracle.ransac(x[off[k]:off[k + 1]], y[off[k]:off[k + 1]], K, S[k], max_sq_err=thr, success_prob=0.999, adaptive=adaptive,
                          loss_kind=loss)
        assert best["best_h"][k] == e["best_h"] and best["iterations"][k] == e["iterations"], (k, n)
        assert np.array_equal(inl[ioff[k]:ioff[k + 1]], e["inliers"]), (k, n)
        if e["best_h"] >= 0:
            assert best["count"][k] == e["counts"][e["best_h"]], (k, n)
            assert np.float64(best["toterr"][k]).view(np.uint64) == np.float64(e["toterr"][e["best_h"]]).view(np.uint64), (k, n)
