"""Matching -> correspondences -> RANSAC with everything resident on the device, on a stand-in for
the reference's default run (main.cpp:49-58: 27 frames, consecutive pairs, threshold 0.95, pixel
gate 100^2, maxSquaredError 1, successProb 0.999), against the CPU restatement stage by stage.
Bar: bit-exact match lists, bit-exact normalised correspondences (scene.cpp:495-501), identical
sample tables, identical inlier lists and per-pair best (count, total error)."""
import numpy as np
import pytest

from sfm_amd import capi, synth

pytestmark = pytest.mark.gpu

N_FRAMES, N_FEAT, N_HYP, SEED = 27, 1500, 1024, 20261111


@pytest.fixture(scope="module")
def sequence():
    return synth.frame_sequence(N_FRAMES, N_FEAT, seed=99)


def test_default_run_stand_in(ctx, oracle, sequence):
    desc, pts, counts, poses = sequence
    K = synth.default_K()
    pl = np.stack([np.arange(N_FRAMES - 1), np.arange(1, N_FRAMES)], axis=1).astype(np.int32)
    s = ctx.imageset(counts, desc, pts)
    res = s.match(pl, threshold=0.95, max_dist_sq=100.0 * 100.0)
    off, pairs = res.offsets(), res.pairs()
    corr = capi.Correspondences(ctx, s, res, K)
    assert np.array_equal(corr.offsets(), off)
    x, y = corr.fetch()
    samples = corr.sample_table(N_HYP, SEED)
    batch = corr.ransac(K, N_HYP, SEED, max_sq_err=1.0, success_prob=0.999, adaptive=True)
    best = batch.best()
    ioff, inl = batch.inliers()
    rpose, rused, rpts = batch.refit(len(x))

    for k, (a, b) in enumerate(pl):
        A, B = slice(a * N_FEAT, (a + 1) * N_FEAT), slice(b * N_FEAT, (b + 1) * N_FEAT)
        exp, _, _ = oracle.match(desc[A], pts[A], desc[B], pts[B], threshold=0.95, max_dist_sq=100.0 * 100.0)
        m = pairs[off[k]:off[k + 1]]
        assert np.array_equal(m, exp), f"pair {k}: match list"
        assert len(m) > 0.4 * N_FEAT
        ex = oracle.normalize_points(K, pts[A][m[:, 0]])
        ey = oracle.normalize_points(K, pts[B][m[:, 1]])
        assert np.array_equal(x[off[k]:off[k + 1]], ex) and np.array_equal(y[off[k]:off[k + 1]], ey), f"pair {k}: rays"
        es = oracle.sample_table(len(m), N_HYP, SEED, k)
        assert np.array_equal(samples[k], es), f"pair {k}: sample table"
        assert np.all(np.sort(es, axis=1)[:, 1:] != np.sort(es, axis=1)[:, :-1]) and es.min() >= 0 and es.max() < len(m)
        r = oracle.ransac(ex, ey, K, es, max_sq_err=1.0, success_prob=0.999, adaptive=True)
        assert best["best_h"][k] == r["best_h"] and best["count"][k] == len(r["inliers"]), f"pair {k}: best hypothesis"
        assert best["toterr"][k] == r["toterr"][r["best_h"]], f"pair {k}: total error"
        assert np.array_equal(inl[ioff[k]:ioff[k + 1]], r["inliers"]), f"pair {k}: inliers"
        # the planted geometry is found: most matches are inliers, and the rotation is the planted one
        assert len(r["inliers"]) > 0.6 * len(m)
        R1, T1 = poses[a]
        R2, T2 = poses[b]
        R_rel = R2 @ R1.T
        Rg = best["pose12"][k][:9].reshape(3, 3)
        ang = np.degrees(np.arccos(np.clip((np.trace(Rg @ R_rel.T) - 1) / 2, -1, 1)))
        assert ang < 0.5, f"pair {k}: rotation off by {ang} deg"
        # refit on the inliers (scene.cpp:288): floating point, a different but equivalent route to the
        # null vector than the restatement's -> tolerance 1e-8 on the pose, 1e-6 relative on the points
        # (one float ulp of the depth, eightPoint.cpp:216, is 6e-8)
        epose, epts, _, _ = oracle.eight_point(ex[r["inliers"]], ey[r["inliers"]])
        assert rused[k] == len(r["inliers"])
        assert np.abs(rpose[k] - epose).max() < 1e-8, f"pair {k}: refit pose"
        got = rpts[off[k]:off[k] + rused[k]]
        assert np.abs(got - epts).max() <= 1e-6 * np.abs(epts).max(), f"pair {k}: refit points"
        ang = np.degrees(np.arccos(np.clip((np.trace(rpose[k][:3, :3] @ R_rel.T) - 1) / 2, -1, 1)))
        assert ang < 0.2, f"pair {k}: refit rotation off by {ang} deg"
    s.close()


def test_sample_table_small_counts(ctx, oracle):
    """Pairs with fewer than 8 matches get zero rows; n == 8 rows are permutations of range(8)."""
    counts = np.array([8, 8, 5, 5], np.int32)
    rng = np.random.default_rng(5)
    desc = np.zeros((26, 128), np.float32)
    base8 = synth.sift_like(rng, 8)
    base5 = synth.sift_like(rng, 5)
    desc[0:8], desc[8:16], desc[16:21], desc[21:26] = base8, base8, base5, base5
    pts = rng.uniform(0, 1000, (26, 2))
    pts[8:16] = pts[0:8] + 1.0
    pts[21:26] = pts[16:21] + 1.0
    s = ctx.imageset(counts, desc, pts)
    res = s.match(np.array([[0, 1], [2, 3]], np.int32), threshold=0.95)
    assert list(np.diff(res.offsets())) == [8, 5]
    corr = capi.Correspondences(ctx, s, res, synth.default_K())
    t = corr.sample_table(64, 7)
    assert np.array_equal(np.sort(t[0], axis=1), np.tile(np.arange(8), (64, 1)))
    assert np.array_equal(t[0], oracle.sample_table(8, 64, 7, 0))
    assert not t[1].any()
    b = corr.ransac(synth.default_K(), 64, 7)
    assert b.best()["count"][1] == 0  # RANSAC.cpp:148-149: fewer than minN correspondences, no result
    s.close()


def test_refit_falls_back_to_all_points(ctx, oracle):
    """Fewer than 8 inliers: the refit uses every correspondence (scene.cpp:294-298); fewer than 8
    correspondences: identity pose, no points."""
    tv = synth.two_view(40, inlier_ratio=1.0, seed=3, noise_px=0.0)
    junk = np.random.default_rng(1).uniform(-0.5, 0.5, (12, 2))
    x = np.concatenate([tv["x"], junk, tv["x"][:5]])
    y = np.concatenate([tv["y"], junk[::-1], tv["y"][:5]])
    off = np.array([0, 40, 52, 57], np.int64)
    samples = np.zeros((3, 4, 8), np.int32)
    samples[0] = synth.sample_table(40, 4, seed=1)
    samples[1] = synth.sample_table(12, 4, seed=2)
    b = ctx.ransac_batch(off, x, y, tv["K"], samples, 4, max_sq_err=1e-9)
    cnt = b.best()["count"]
    assert cnt[0] >= 8 and cnt[1] < 8 and cnt[2] == 0
    pose, used, pts = b.refit(len(x))
    assert list(used) == [cnt[0], 12, 0]
    _, inl = b.inliers()
    e0, p0, _, _ = oracle.eight_point(x[:40][inl[:cnt[0]]], y[:40][inl[:cnt[0]]])
    e1, p1, _, _ = oracle.eight_point(x[40:52], y[40:52])
    assert np.abs(pose[0] - e0).max() < 1e-8 and np.abs(pose[1] - e1).max() < 1e-6
    assert np.array_equal(pose[2], np.eye(4))
    assert np.abs(pts[:cnt[0]] - p0).max() <= 1e-6 * np.abs(p0).max()


@pytest.mark.parametrize("n", [8, 9, 100, 3000])
def test_eight_point_algorithm(ctx, oracle, n):
    tv = synth.two_view(n, inlier_ratio=1.0, seed=n, noise_px=0.2)
    pose, pts = ctx.eight_point_algorithm(tv["x"], tv["y"])
    epose, epts, _, _ = oracle.eight_point(tv["x"], tv["y"])
    tol = 1e-8 if n > 9 else 1e-5  # 8 or 9 points: the fit is exact or nearly so, hence ill-conditioned in the noise
    assert np.abs(pose - epose).max() < tol
    assert np.abs(pts - epts).max() <= max(1e-6, 10 * tol) * np.abs(epts).max()
    if n >= 100:
        ang = np.degrees(np.arccos(np.clip((np.trace(pose[:3, :3] @ tv["R"].T) - 1) / 2, -1, 1)))
        assert ang < 1.0


def test_eight_point_algorithm_needs_eight_points(ctx):
    with pytest.raises(capi.SfmError) as e:
        ctx.eight_point_algorithm(np.zeros((7, 2)), np.zeros((7, 2)))
    assert e.value.code == capi.ERR_INVALID_ARG and "at least 8" in str(e.value)
