"""The HIP path (through the C-ABI) against the committed golden vectors in tests/golden/."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "match_*.npz"))), ids=os.path.basename)
def test_matcher_golden(ctx, path):
    g = np.load(path)
    pairs, m1, m2 = ctx.match_pair(g["descA"], g["ptsA"], g["descB"], g["ptsB"], threshold=float(g["threshold"]),
                                   max_dist_sq=float(g["max_dist_sq"]), metric=int(g["metric"]), return_best=True)
    assert np.array_equal(pairs, g["pairs"]) and np.array_equal(m1, g["matches1"]) and np.array_equal(m2, g["matches2"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "ransac_*.npz"))), ids=os.path.basename)
def test_ransac_golden(ctx, path):
    g = np.load(path)
    r = ctx.ransac_pair(g["x"], g["y"], g["K"], g["samples"], max_sq_err=float(g["max_sq_err"]),
                        success_prob=float(g["success_prob"]), adaptive=bool(g["adaptive"]))
    assert np.array_equal(r["inliers"], g["inliers"]) and r["best_h"] == int(g["best_h"])
    assert np.array_equal(r["counts"], g["counts"]) and r["iterations"] == int(g["iterations"])
    assert np.allclose(r["toterr"], g["toterr"], rtol=1e-12, atol=0)
    assert np.abs(r["pose"] - g["pose"]).max() <= 1e-6
    pose12 = ctx.eight_point_fit(g["x"], g["y"], g["samples"])
    assert np.abs(pose12 - g["pose12"]).max() <= 1e-6


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "ratio_*.npz"))), ids=os.path.basename)
def test_ratio_golden(ctx, path):
    g = np.load(path)
    nA, nB = len(g["ptsA"]), len(g["ptsB"])
    s = ctx.imageset(np.array([nA, nB], np.int32), np.concatenate([g["descA"], g["descB"]]),
                     np.concatenate([g["ptsA"], g["ptsB"]]))
    res = s.match(np.array([[0, 1]], np.int32), threshold=float(g["threshold"]), max_dist_sq=float(g["max_dist_sq"]),
                  ratio=float(g["ratio"]))
    assert np.array_equal(res.pairs(), g["pairs"])
    res.close()
    s.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "l2_*.npz"))), ids=os.path.basename)
def test_l2_golden(ctx, path):
    from sfm_amd import capi
    g = np.load(path)
    nA, nB = len(g["ptsA"]), len(g["ptsB"])
    s = ctx.imageset(np.array([nA, nB], np.int32), np.concatenate([g["descA"], g["descB"]]),
                     np.concatenate([g["ptsA"], g["ptsB"]]))
    res = s.match(np.array([[0, 1]], np.int32), threshold=float(g["max_distance_squared"]), max_dist_sq=float(g["max_dist_sq"]),
                  metric=capi.METRIC_L2, ratio=float(g["ratio"]))
    assert np.array_equal(res.pairs(), g["pairs"])
    res.close()
    s.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "hamming_*.npz"))), ids=os.path.basename)
def test_hamming_golden(ctx, path):
    from sfm_amd import capi
    g = np.load(path)
    nA, nB = len(g["ptsA"]), len(g["ptsB"])
    s = capi.BinarySet(ctx, np.array([nA, nB], np.int32), np.concatenate([g["descA"], g["descB"]]),
                       np.concatenate([g["ptsA"], g["ptsB"]]))
    res = s.match(np.array([[0, 1]], np.int32), max_distance=int(g["max_distance"]), max_dist_sq=float(g["max_dist_sq"]))
    assert np.array_equal(res.pairs(), g["pairs"])
    res.close()
    s.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "chain_*.npz"))), ids=os.path.basename)
def test_chain_golden(ctx, path):
    g = np.load(path)
    r = ctx.scale_chain(g["offsets"], g["track_ids"], g["points"], g["poses"], start_pose=g["start"],
                        max_translation=float(g["max_translation"]), n_tracks=int(g["n_tracks"]))
    for k in ("extrinsics", "points3d", "counts", "scale"):
        assert np.array_equal(r[k], g[k]), k


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "refit_*.npz"))), ids=os.path.basename)
def test_refit_golden(ctx, path):
    """eightPointAlgorithm (eightPoint.cpp:57-223): the device reaches the null vector through AtA, the
    restatement through A itself: poses to 1e-8, points to 1e-6 relative."""
    g = np.load(path)
    pose, points = ctx.eight_point_algorithm(g["x"], g["y"])
    assert np.abs(pose - g["pose"]).max() <= 1e-8
    assert np.abs(points - g["points"]).max() <= 1e-6 * np.abs(g["points"]).max()
