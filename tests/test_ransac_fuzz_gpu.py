"""Randomised sweep of RANSAC scoring against the CPU restatement (reference submodules/solve/RANSAC.cpp:146-270,
eightPoint.cpp:225-372, solve.hpp:85-124): correspondence counts, inlier ratios, noise, intrinsics, error
bounds and sample tables drawn from a seeded generator.  Bar: identical per-row inlier counts, bit-identical
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
    ctx.set_option("ransac_filter", int(rng.integers(2)))
    try:
        got = ctx.ransac_pair(x, y, K, S, max_sq_err=thr, success_prob=0.999, adaptive=adaptive)
    finally:
        ctx.set_option("ransac_filter", 1)
    exp = oracle.ransac(x, y, K, S, max_sq_err=thr, success_prob=0.999, adaptive=adaptive)
    assert np.array_equal(got["counts"], exp["counts"]), "per-row inlier counts differ"
    assert np.array_equal(got["toterr"].view(np.uint64), exp["toterr"].view(np.uint64)), "per-row error sums differ"
    assert got["best_h"] == exp["best_h"] and got["iterations"] == exp["iterations"]
    assert np.array_equal(got["inliers"], exp["inliers"])
